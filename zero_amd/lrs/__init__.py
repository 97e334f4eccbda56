# coding: utf-8
"""Learning-rate schedules fed to the step as a host scalar (main.py:157,280,292).

Only what the hot path needs: ``noam`` (lrs/noamlr.py:27-34) and a constant rate, with the
clamp of lrs/lr.py:43-45.  The remaining schedules of the reference (gnmt+, epoch, score,
cosine) are host-side control plane and out of scope."""


class Lr(object):
    def __init__(self, init_lrate, min_lrate, max_lrate, name="lr"):
        assert max_lrate > min_lrate, "Minimum learning rate should less than maximum learning rate"
        self.name = name
        self.init_lrate = init_lrate
        self.lrate = init_lrate
        self.min_lrate = min_lrate
        self.max_lrate = max_lrate

    def step(self, step):
        pass

    def get_lr(self):
        return max(min(self.lrate, self.max_lrate), self.min_lrate)


class NoamDecayLr(Lr):
    def __init__(self, init_lr, min_lr, max_lr, warmup_steps, hidden_size, name="noam_decay_lr"):
        super(NoamDecayLr, self).__init__(init_lr, min_lr, max_lr, name=name)
        self.warmup_steps = warmup_steps
        self.hidden_size = hidden_size

    def step(self, step):
        step = float(step)
        w = float(self.warmup_steps)
        decay = float(self.hidden_size) ** -0.5 * min((step + 1) * (w ** -1.5), (step + 1) ** -0.5)
        self.lrate = self.init_lrate * decay


def get_lr(params):
    strategy = params.lrate_strategy.lower()
    if strategy == "noam":
        return NoamDecayLr(params.lrate, params.min_lrate, params.max_lrate, params.warmup_steps,
                           params.hidden_size)
    if strategy in ("vanilla", "constant"):
        return Lr(params.lrate, params.min_lrate, params.max_lrate)
    raise NotImplementedError("lrate_strategy %r is host-side control plane outside the hot path; "
                              "use 'noam' or 'vanilla'" % params.lrate_strategy)
