# coding: utf-8
"""Training-progress record saved beside the checkpoints (utils/recorder.py:11-24; fields set in
run.py:275-293): a plain attribute bag dumped to / loaded from ``record.json`` (indent 2)."""

import json


class Recorder(object):
    def load_from_json(self, file_name):
        with open(file_name, "r") as reader:
            self.__dict__.update(json.load(reader))

    def save_to_json(self, file_name):
        with open(file_name, "w") as writer:
            writer.write(json.dumps(self.__dict__, indent=2))


def new_recorder(params):
    """The fields run.py:275-289 initialises before training."""
    r = Recorder()
    r.bad_counter = 0
    r.estop = False
    r.lidx = -1
    r.step = 0
    r.epoch = 1
    r.lrate = params.lrate
    r.history_scores = []
    r.valid_script_scores = []
    return r
