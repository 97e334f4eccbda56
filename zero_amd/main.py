# coding: utf-8
"""The step section of the reference's main.py on the HIP path.

Counterparts:
  * ``tower_train_graph`` (main.py:22-45)  -> :func:`tower_train_graph`: one tower per
    process; ``train_fn`` -> loss + gradients; the tower average is the RCCL all-reduce of
    zero_amd/utils/parallel.py, launched bucket by bucket during the backward.
  * the hot loop of ``train`` (main.py:268-332) -> :class:`Trainer`: zero_op at cycle start,
    collect_op for the first update_cycle-1 micro steps, train_op on the last, LR fed as a
    host scalar evaluated before the step (main.py:280; lrs/noamlr.py).
  * ``tower_infer_graph`` (main.py:48-62) / ``tower_score_graph`` (main.py:118-130)
    -> :func:`tower_infer_graph` / :func:`tower_score_graph`.

Data loading, evaluation cadence, checkpoints and logging of main.py are host control plane
outside the hot path.
"""

import torch

from zero_amd import lrs
from zero_amd.models import model as model_registry
from zero_amd.models import load_all
from zero_amd.utils import parallel
from zero_amd.utils.cycle import TrainOp


def tower_train_graph(train_features, graph, params, reducer=None):
    """main.py:22-45 for this rank's tower.  Returns (loss tensor, train_fn output)."""
    out = graph.train_fn(train_features, params, on_ready=reducer.ready if reducer is not None else None)
    return out["loss"], out


def tower_score_graph(eval_features, graph, params):
    """main.py:118-130."""
    return graph.score_fn(eval_features, params)["score"]


def tower_infer_graph(eval_features, graph, params):
    """main.py:48-62."""
    from zero_amd.search import beam_search
    encoding_fn, decoding_fn = graph.infer_fn(params)
    out = beam_search(eval_features, encoding_fn, decoding_fn, params)
    return out["seq"], out["score"]


class Trainer(object):
    """One data-parallel replica of the training step (main.py:255-332)."""

    def __init__(self, params, initializer=None):
        load_all()
        self.params = params
        self.graph = model_registry.get_model(params.model_name)
        from zero_amd.models._factory import get_core
        self.core = get_core(params, params.model_name, initializer)
        self.store = self.core.store
        self.train_op = TrainOp(self.store, params, self.core.eng)
        self.reducer = parallel.GradientAllReduce(self.store)
        self.lr = lrs.get_lr(params)
        self.global_step = 0
        self.cycle_counter = 0
        self._graphs = {}

    # -- eager path (any shapes) --------------------------------------------------
    def micro_step(self, features):
        """One forward+backward (+ update on the last micro step of a cycle).
        Returns the loss tensor (device, not synchronised)."""
        hp = self.params
        world = parallel.world_size()
        if self.cycle_counter == 0:
            self.train_op.zero()
        last = (self.cycle_counter + 1) >= hp.update_cycle
        self.lr.step(self.global_step)
        overlap = last and hp.update_cycle == 1
        loss, _ = tower_train_graph(features, self.graph, hp, self.reducer if overlap else None)
        if not last:
            self.train_op.collect()
            self.cycle_counter += 1
            return loss
        if hp.update_cycle > 1:
            # reference semantics: one average over N*c micro batches (cycle.py:86-88)
            scale = self.train_op.apply(self.lr.get_lr(), world, launch=False)
            self.reducer.all_reduce_everything()
            self.train_op.launch_update(scale)
        else:
            self.reducer.wait()
            self.train_op.apply(self.lr.get_lr(), world)
        self.cycle_counter = 0
        self.global_step += 1
        self.core.eng.lib.call("zk_seed_advance", self.core.eng.seed.data_ptr(), 1, self.core.eng.stream)
        return loss

    # -- captured path (static shapes, update_cycle == 1) ---------------------------
    def prepare_static(self, features):
        """Upload one batch into the static id buffers; later steps may overwrite the
        same buffers (same shapes) with :meth:`refill`."""
        self.batch = self.core.upload(features["source"], features["target"])
        return self.batch

    def step_static(self, use_graph=True):
        """One full update on the static batch: fwd + bwd (+ all-reduce) + Adam.
        With a single rank the whole step is one hipGraph replay."""
        eng = self.core.eng
        cur = torch.cuda.current_stream(eng.device)
        ws = eng.work_stream
        ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            loss = self._step_static(use_graph)
        cur.wait_stream(ws)
        return loss

    def _step_static(self, use_graph):
        hp = self.params
        assert hp.update_cycle == 1, "captured step supports update_cycle == 1"
        world = parallel.world_size()
        eng = self.core.eng
        self.lr.step(self.global_step)
        self.train_op.count = 0
        scale = self.train_op.set_hyper(self.lr.get_lr(), world)
        if world == 1 and use_graph:
            key = (self.batch["B"], self.batch["Ls"], self.batch["Lt"])
            g = self._graphs.get(key)
            if g is None:
                # first use of a shape runs eagerly once (sizes every scratch buffer), then
                # the same launch sequence is captured
                self._graphs[key] = "warm"
                return self._step_static(False)
            if g == "warm":
                def body():
                    self.graph.train_fn(self.batch, hp)
                    self.train_op.launch_update(scale)
                    eng.lib.call("zk_seed_advance", eng.seed.data_ptr(), 1, eng.stream)
                g = eng.graph_capture(body)
                self._graphs[key] = g
            eng.graph_launch(g)
        else:
            self.graph.train_fn(self.batch, hp, on_ready=self.reducer.ready if world > 1 else None)
            self.reducer.wait()
            self.train_op.launch_update(scale)
            eng.lib.call("zk_seed_advance", eng.seed.data_ptr(), 1, eng.stream)
        self.store.step += 1
        self.global_step += 1
        return eng.buf("loss", (1,), torch.float32)
