# coding: utf-8
"""Evaluation loop around the decode / score hot paths (evalu.py of the reference;
SURVEY.md §8(f)-4).  The TF session + placeholder plumbing is gone: ``decoding`` and ``scoring``
take the model wrapper (models/model.py) and call ``tower_infer_graph`` / ``tower_score_graph``
of zero_amd/main.py on each batch.

  decode_target_token  evalu.py:14-22   cut at the first eos **or pad**, map ids to tokens
  decode_hypothesis    evalu.py:25-46   beam 0 of every sentence (+ its score)
  decoding             evalu.py:49-139  batches in length-sorted order; returns
                                        (translations, scores, indices) in *batch* order
  scoring              evalu.py:142-243 per-sentence scores restored to file order, and
                                        ppl = exp(sum_s score_s * len_s / sum_s len_s)
  eval_metric          evalu.py:246-263 BLEU against ``target_file`` (or ``target_file0..``)
  dump_tanslation      evalu.py:266-280 one line per hypothesis, optionally re-ordered by index
"""

import logging
import os
import time

import numpy as np

from zero_amd.utils import metric, queuer

log = logging.getLogger("zero_amd")


def decode_target_token(id_seq, vocab):
    keep = []
    for tok in id_seq:
        tok = int(tok)
        if tok == vocab.eos() or tok == vocab.pad():
            break
        keep.append(tok)
    return vocab.to_tokens(keep)


def decode_hypothesis(seqs, scores, params, mask=None):
    """``seqs`` / ``scores``: per tower, [B, K, L] ids and [B, K] scores."""
    if mask is None:
        mask = [1.] * len(seqs)
    hypoes, marks = [], []
    for tower_seqs, tower_scores, m in zip(seqs, scores, mask):
        if m < 1.:
            continue
        for seq, score in zip(tower_seqs, tower_scores):
            hypoes.append(decode_target_token(seq[0], params.tgt_vocab))
            marks.append(score[0])
    return hypoes, marks


def _batches(dataset, params):
    return queuer.EnQueuer(
        dataset.batcher(params.eval_batch_size, buffer_size=params.buffer_size, shuffle=False, train=False),
        lambda x: x,
        worker_processes_num=params.process_num,
        input_queue_size=params.input_queue_size,
        output_queue_size=params.output_queue_size,
    )


def decode_streams(default=4):
    """How many decode batches are kept in flight at once (ZERO_HIP_DECODE_STREAMS; 1 = one after the other)."""
    try:
        return max(1, min(8, int(os.environ.get("ZERO_HIP_DECODE_STREAMS", str(default)))))
    except ValueError:
        return default


_HWQ_WARNED = []


def _warn_hw_queues(streams):
    try:
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return
    if int(streams) > 1 and q < int(streams) + 2 and not _HWQ_WARNED:
        _HWQ_WARNED.append(1)
        log.warning("decode_many: %d batches in flight on GPU_MAX_HW_QUEUES=%d hardware queues -- lanes that share a "
                    "queue serialise; export GPU_MAX_HW_QUEUES=8 before the first HIP call of the process", int(streams), q)


def decode_many(items, work, streams=None, each_lane=False):
    """``[work(item) for item in items]`` with up to ``streams`` items in flight at once, each on its own execution
    lane (zero_amd.models._factory.lane: own HIP stream, scratch and cache buffers, captured step graphs; the variable
    store is shared).  Results come back in the order of ``items``.

    Why: one beam-search decode step works on eval_batch_size x beam rows (32 x 4 = 128) -- ~40 dependent launches of
    which none fills a quarter of the 256 CUs; the step is a latency chain.  Independent batches
    have no data in common but the weights, so their chains interleave on the device for nothing: the reference's
    evaluation loop (evalu.py:49-139) decodes its batches one after the other only because a TF session runs one
    ``session.run`` at a time.  Every batch's arithmetic is exactly what it is alone (same kernels, same buffers
    per lane), so hypotheses and scores are bit-identical to the sequential loop (tests/test_gpu_model.py).
    each_lane=True: EVERY lane works through all items (sizing its buffers and its graph cache: the warm-up of a
    measurement); the results of lane 0 are returned.

    Host side: one worker thread per lane (ctypes calls and stream / event waits release the GIL; the Python work
    per decode step is ~30 us against ~360 us of device time).

    Hardware queues: HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4),
    round-robin; two lanes that share a queue run one after the other again (4 lanes: 2740 sentences/s on 4 queues,
    3340 on 8, profiles/r03_bench_decode_models.jsonl).  The runtime reads the variable once, at its first HIP call,
    so it is the HOST PROGRAM's to set before that (``zero_amd.run`` and ``bench.py`` export GPU_MAX_HW_QUEUES=8 at
    start-up unless the user chose a value); a library import must not change the process environment behind the
    host's back, so this function only warns when it finds fewer queues than lanes + 2."""
    _warn_hw_queues(decode_streams() if streams is None else streams)
    import threading
    streams = decode_streams() if streams is None else max(1, int(streams))
    if each_lane:
        items = list(items)
    it = iter(items)
    if streams == 1:
        return [work(x) for x in it]
    import torch
    from zero_amd.models._factory import lane
    from zero_amd.models._decode import lanes_mode
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    if dev is not None:
        # Whatever the CALLER has enqueued on its current stream -- an EMA weight swap (main._evaluate_dev: ema_assign's
        # master copy + shadow refresh run on the training loop's work stream), a checkpoint load -- must be complete
        # before any lane reads the shared variable store: a lane's stream only orders itself behind its OWN thread's
        # current stream, which is the null stream in a fresh thread.  One host wait per evaluation.
        torch.cuda.current_stream(dev).synchronize()
    lock = threading.Lock()
    results, errors = {}, []
    counter = [0]

    def worker(idx):
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            with lane(idx):
                if each_lane:
                    out = [work(x) for x in items]
                    if idx == 0:
                        results.update(enumerate(out))
                        counter[0] = len(out)
                    return
                while not errors:
                    with lock:
                        try:
                            x = next(it)
                        except StopIteration:
                            return
                        k = counter[0]
                        counter[0] += 1
                    results[k] = work(x)
        except BaseException as exc:      # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,), name="decode-lane-%d" % i) for i in range(streams)]
    with lanes_mode(streams):
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return [results[k] for k in range(counter[0])]


def decoding(graph, dataset, params, infer=None):
    """Translate ``dataset``; ``infer(features, graph, params) -> (seqs [B,K,L], scores [B,K])``
    defaults to zero_amd.main.tower_infer_graph.  Batches are decoded ``decode_streams()`` at a time (decode_many)."""
    if infer is None:
        from zero_amd.main import tower_infer_graph as infer
    translations, scores, indices = [], [], []
    begin = time.time()

    def work(data):
        start = time.time()
        seqs, sc = infer({"source": data['src']}, graph, params)
        return data, np.asarray(seqs), np.asarray(sc), time.time() - start

    # (dev-mode search re-encodes on the training path of lane 0's engine: one batch at a time)
    streams = decode_streams() if getattr(params, "search_mode", "cache") == "cache" else 1
    for bidx, (data, seqs, sc, used) in enumerate(decode_many(_batches(dataset, params), work, streams)):
        hyp, marks = decode_hypothesis([seqs], [sc], params)
        translations.extend(hyp)
        scores.extend(float(m) for m in marks)
        indices.extend(data['index'])
        log.info("Decoding Batch %s using %.3f s, translating %d sentences using %.3f s in total",
                 bidx, used, len(translations), time.time() - begin)
    return translations, scores, indices


def scoring(graph, dataset, params, score=None):
    """Per-sentence length-normalised losses in file order and corpus perplexity."""
    if score is None:
        from zero_amd.main import tower_score_graph as score
    scores, indices = [], []
    total_entropy = total_tokens = 0.
    for bidx, data in enumerate(_batches(dataset, params)):
        s = _to_numpy(score({"source": data['src'], "target": data['tgt']}, graph, params))
        lens = (data['tgt'] > 0).sum(axis=1)
        total_entropy += float(sum(si * float(li) for si, li in zip(s.tolist(), lens)))
        total_tokens += float(lens.sum())
        scores.extend(s.tolist())
        indices.extend(data['index'])
    scores = [d[1] for d in sorted(zip(indices, scores), key=lambda x: x[0])]
    ppl = np.exp(total_entropy / total_tokens)
    return scores, ppl


def _to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().float().cpu().numpy()
    return np.asarray(x)


def fetch_valid_ref_files(path):
    """utils/util.py:232-251: ``path`` itself, else ``path.ref0, path.ref1, ...``; None (with a
    warning) if neither exists."""
    path = os.path.abspath(path)
    if os.path.exists(path):
        return [path]
    files = []
    while os.path.exists(path + ".ref%s" % len(files)):
        files.append(path + ".ref%s" % len(files))
    if not files:
        log.warning("Invalid Reference Format %s", path)
        return None
    return files


def eval_metric(trans, target_file, indices=None):
    files = fetch_valid_ref_files(target_file)
    if files is None:
        return 0.0
    if indices is not None:
        trans = [d[1] for d in sorted(zip(indices, trans), key=lambda x: x[0])]
    references = [[line.strip().split() for line in open(f).readlines()] for f in files]
    return metric.bleu(trans, list(zip(*references)))


def dump_tanslation(tranes, output, indices=None):
    if indices is not None:
        tranes = [d[1] for d in sorted(zip(indices, tranes), key=lambda x: x[0])]
    os.makedirs(os.path.dirname(os.path.abspath(output)), exist_ok=True)
    with open(output, 'w') as writer:
        for hypo in tranes:
            writer.write((' '.join(hypo) if isinstance(hypo, list) else str(hypo)) + "\n")
    log.info("Saving translations into %s", output)
