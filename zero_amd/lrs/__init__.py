# coding: utf-8
"""Learning-rate schedules fed to the step as a host scalar (main.py:157,280,292).

All six strategies of the reference's ``lrs`` package behind the same ``get_lr(params)``
factory (lrs/__init__.py:6-62) and the same life-cycle hooks (lrs/lr.py:30-45):
``before_epoch``, ``after_epoch``, ``step``, ``after_eval``, ``get_lr`` (clamped to
``[min_lrate, max_lrate]``).  Pinned against values produced by importing the reference's
own (TF-free) ``lrs`` modules: tests/golden/reference_host.json.
"""

import math


class Lr(object):
    """lrs/lr.py:14-45; also the ``vanilla`` (constant) strategy, lrs/vanillalr.py."""

    def __init__(self, init_lrate, min_lrate, max_lrate, name="lr"):
        assert max_lrate > min_lrate, "Minimum learning rate should less than maximum learning rate"
        self.name = name
        self.init_lrate = init_lrate
        self.lrate = init_lrate
        self.min_lrate = min_lrate
        self.max_lrate = max_lrate

    def before_epoch(self, eidx=None):
        pass

    def after_epoch(self, eidx=None):
        pass

    def step(self, step):
        pass

    def after_eval(self, eval_score):
        pass

    def get_lr(self):
        return max(min(self.lrate, self.max_lrate), self.min_lrate)


VanillaLR = Lr


class NoamDecayLr(Lr):
    """lrs/noamlr.py:27-34: ``init * H^-0.5 * min((s+1) w^-1.5, (s+1)^-0.5)``."""

    def __init__(self, init_lr, min_lr, max_lr, warmup_steps, hidden_size, name="noam_decay_lr"):
        super(NoamDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        self.warmup_steps = warmup_steps
        self.hidden_size = hidden_size

    def step(self, step):
        step = float(step)
        w = float(self.warmup_steps)
        decay = float(self.hidden_size) ** -0.5 * min((step + 1) * (w ** -1.5), (step + 1) ** -0.5)
        self.lrate = self.init_lrate * decay


class GNMTPDecayLr(Lr):
    """lrs/gnmtplr.py:35-45: linear warm-up to ``n`` replicas' rate over ``p`` steps, flat, then
    exponential decay ``n (2n)^((s - n t)/(e - s))`` between ``lrdecay_start`` and ``_end``."""

    def __init__(self, init_lr, min_lr, max_lr, warmup_steps, nstable, lrdecay_start, lrdecay_end,
                 name="gnmtp_decay_lr"):
        super(GNMTPDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        if nstable < 1:
            raise Exception("Stabled Lrate Value should greater than 0, but is {}".format(nstable))
        self.warmup_steps = warmup_steps
        self.nstable = nstable
        self.lrdecay_start = lrdecay_start
        self.lrdecay_end = lrdecay_end

    def step(self, step):
        t, p, n = float(step), float(self.warmup_steps), float(self.nstable)
        s, e = float(self.lrdecay_start), float(self.lrdecay_end)
        warm = min(1.0 + t * (n - 1.0) / (n * p), n)
        tail = n * (2.0 * n) ** ((s - n * t) / (e - s))
        self.lrate = self.init_lrate * min(warm, tail)


class EpochDecayLr(Lr):
    """lrs/epochlr.py:23-28: ``init * decay^epoch`` set after each epoch."""

    def __init__(self, init_lr, min_lr, max_lr, decay=0.5, name="epoch_decay_lr"):
        super(EpochDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        self.decay = decay

    def after_epoch(self, eidx=None):
        self.lrate = self.init_lrate * self.decay ** (1 if eidx is None else int(eidx))


class ScoreDecayLr(Lr):
    """lrs/scorelr.py:30-42: multiply by ``decay`` after ``patience`` evaluations without a new
    best score.  ``history_scores`` replays earlier evaluations; entries may be plain scores or
    ``(step, score)`` pairs as the recorder stores them (main.py:397) -- the reference indexes
    ``score[1]`` of what its factory already reduced to floats (lrs/__init__.py:36 vs
    scorelr.py:28), which raises on a non-empty history; both forms are accepted here."""

    def __init__(self, init_lr, min_lr, max_lr, history_scores=None, decay=0.5, patience=1,
                 name="score_decay_lr"):
        super(ScoreDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        self.decay = decay
        self.patience = patience
        self.bad_counter = 0
        self.best_score = -1e9
        for score in history_scores or []:
            self.after_eval(score[1] if isinstance(score, (tuple, list)) else score)

    def after_eval(self, eval_score):
        if eval_score > self.best_score:
            self.best_score = eval_score
            self.bad_counter = 0
            return
        self.bad_counter += 1
        if self.bad_counter >= self.patience:
            self.lrate = self.lrate * self.decay
            self.bad_counter = 0


class CosineDecayLr(Lr):
    """lrs/cosinelr.py:44-63 (fairseq-style): linear warm-up from ``init`` to ``max`` over
    ``warmup_steps``, then cosine annealing between ``max`` and ``min`` in periods of
    ``update_period * t_mult^i`` updates, both ends shrunk by ``decay^i``."""

    def __init__(self, init_lr, min_lr, max_lr, warmup_steps, decay, t_mult=1, update_period=5000,
                 name="cosine_decay_lr"):
        super(CosineDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        self.warmup_steps = warmup_steps
        self.t_mult = t_mult
        self.period = update_period
        self.decay = decay
        self.lr_step = (max_lr - init_lr) / warmup_steps if warmup_steps > 0 else 1.0

    def step(self, step):
        if step < self.warmup_steps:
            self.lrate = self.init_lrate + step * self.lr_step
            return self.lrate
        done = step - self.warmup_steps
        if self.t_mult != 1:
            i = math.floor(math.log(1 - done / self.period * (1 - self.t_mult), self.t_mult))
            span = self.t_mult ** i * self.period
            pos = done - (1 - self.t_mult ** i) / (1 - self.t_mult) * self.period
        else:
            i = math.floor(done / self.period)
            span = self.period
            pos = done - self.period * i
        shrink = self.decay ** i
        lo, hi = self.min_lrate * shrink, self.max_lrate * shrink
        self.lrate = lo + 0.5 * (hi - lo) * (1 + math.cos(math.pi * pos / span))
        return self.lrate


def get_lr(params):
    """lrs/__init__.py:6-62."""
    strategy = params.lrate_strategy.lower()
    base = (params.lrate, params.min_lrate, params.max_lrate)
    if strategy == "noam":
        return NoamDecayLr(*base, warmup_steps=params.warmup_steps, hidden_size=params.hidden_size)
    if strategy == "gnmt+":
        return GNMTPDecayLr(*base, warmup_steps=params.warmup_steps, nstable=params.nstable,
                            lrdecay_start=params.lrdecay_start, lrdecay_end=params.lrdecay_end)
    if strategy == "epoch":
        return EpochDecayLr(*base, decay=params.lrate_decay)
    if strategy == "score":
        recorder = getattr(params, "recorder", None)
        history = list(getattr(recorder, "valid_script_scores", []) or [])
        return ScoreDecayLr(*base, history_scores=history, decay=params.lrate_decay,
                            patience=params.lrate_patience)
    if strategy in ("vanilla", "constant"):
        return Lr(*base)
    if strategy == "cosine":
        return CosineDecayLr(*base, warmup_steps=params.warmup_steps, decay=params.lrate_decay,
                             t_mult=params.cosine_factor, update_period=params.cosine_period)
    raise NotImplementedError("{} is not supported".format(strategy))
