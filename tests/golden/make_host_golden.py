# coding: utf-8
"""Golden vectors for the host rows of SURVEY.md §8(f) (run in the BUILD container only).

``reference_host.json`` holds inputs and outputs of the reference's own TF-free modules, obtained
by IMPORTING them from /root/reference (never copied): every learning-rate schedule of ``lrs/``,
``utils/metric.py`` (bleu / otem / utem on seeded random corpora) and ``utils/queuer.py``
(chunks delivered for worker counts 0, 1 and 3).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    from lrs import cosinelr, epochlr, gnmtplr, noamlr, scorelr, vanillalr   # reference code, imported
    from utils import metric, queuer
    out = {}

    # ---- learning-rate schedules: {name: {"args": [...], "trace": [[event, arg, lr], ...]}}
    def trace(obj, events):
        tr = []
        for ev, arg in events:
            getattr(obj, ev)(arg)
            tr.append([ev, arg, obj.get_lr()])
        return tr
    steps = [("step", s) for s in (0, 1, 7, 399, 400, 401, 4000, 19999, 20000, 123456, 1300000)]
    lr_cases = {}
    lr_cases["noam"] = {"args": [1.0, 1e-5, 1.0, 400, 512], "trace": trace(noamlr.NoamDecayLr(1.0, 1e-5, 1.0, 400, 512), steps)}
    lr_cases["gnmt+"] = {"args": [5e-4, 1e-6, 1e-2, 500, 4, 600000, 1200000],
                         "trace": trace(gnmtplr.GNMTPDecayLr(5e-4, 1e-6, 1e-2, 500, 4, 600000, 1200000),
                                        steps + [("step", s) for s in (150000, 151000, 299999, 300000, 600000)])}
    lr_cases["cosine"] = {"args": [1e-7, 1e-9, 1e-3, 400, 0.5, 1, 5000],
                          "trace": trace(cosinelr.CosineDecayLr(1e-7, 1e-9, 1e-3, 400, 0.5, t_mult=1, update_period=5000), steps)}
    lr_cases["cosine_tmult2"] = {"args": [1e-7, 1e-9, 1e-3, 400, 0.75, 2, 1000],
                                 "trace": trace(cosinelr.CosineDecayLr(1e-7, 1e-9, 1e-3, 400, 0.75, t_mult=2, update_period=1000), steps)}
    lr_cases["epoch"] = {"args": [1.0, 1e-3, 2.0, 0.5],
                         "trace": trace(epochlr.EpochDecayLr(1.0, 1e-3, 2.0, 0.5),
                                        [("after_epoch", None), ("after_epoch", 1), ("after_epoch", 3), ("after_epoch", 20)])}
    lr_cases["score"] = {"args": [1.0, 1e-3, 2.0, 0.5, 2],
                         "trace": trace(scorelr.ScoreDecayLr(1.0, 1e-3, 2.0, decay=0.5, patience=2),
                                        [("after_eval", v) for v in (10.0, 9.0, 8.5, 11.0, 10.0, 10.5, 10.9, 12.0, 1.0, 1.0, 1.0, 1.0)])}
    lr_cases["vanilla"] = {"args": [3.0, 1e-3, 2.0], "trace": trace(vanillalr.VanillaLR(3.0, 1e-3, 2.0), steps[:3])}
    out["lrs"] = lr_cases

    # ---- metric: seeded random corpora with controlled overlap, 1..3 references per sentence
    rnd = random.Random(20260927)
    words = ["w%d" % i for i in range(12)]
    corpora = []
    for case in range(8):
        cand, refs = [], []
        nref = 1 + case % 3
        for _ in range(rnd.randint(1, 9)):
            base = [rnd.choice(words) for _ in range(rnd.randint(0 if case == 5 else 1, 14))]
            c = [w if rnd.random() < 0.7 else rnd.choice(words) for w in base]
            if rnd.random() < 0.3:
                c = c + c[-2:]                  # repeated tail: over-translation
            if rnd.random() < 0.3:
                c = c[:max(1, len(c) // 2)]     # truncated: under-translation
            rs = []
            for _ in range(nref):
                r = [w if rnd.random() < 0.8 else rnd.choice(words) for w in base]
                if rnd.random() < 0.2:
                    r = r + [rnd.choice(words)]
                rs.append(r)
            cand.append(c)
            refs.append(rs)
        rec = {"cand": cand, "refs": refs}
        for bp in ("closest", "shortest"):
            for smooth in (False, True):
                key = "%s_%s" % (bp, "smooth" if smooth else "plain")
                rec[key] = {"bleu": metric.bleu(cand, refs, bp=bp, smooth=smooth),
                            "otem": metric.otem(cand, refs, bp=bp, smooth=smooth),
                            "utem": metric.utem(cand, refs, bp=bp, smooth=smooth)}
        corpora.append(rec)
    out["metric"] = corpora
    out["metric_identity_bleu"] = metric.bleu([["a", "b", "c", "d", "e"]], [[["a", "b", "c", "d", "e"]]])

    # ---- queuer: what an iteration delivers
    def reader():
        for i in range(23):
            yield [i, i * i]
    q = {}
    for n in (0, 1, 3):
        got = list(queuer.EnQueuer(reader(), lambda c: [c[0], c[1] + 1], worker_processes_num=n,
                                   input_queue_size=4, output_queue_size=4))
        q[str(n)] = got if n < 2 else sorted(got)
    out["queuer"] = q
    sys.path.pop(0)
    with open(os.path.join(HERE, "reference_host.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote reference_host.json")


if __name__ == "__main__":
    main()
