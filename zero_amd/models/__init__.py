# coding: utf-8
"""Model package: importing it registers every model module, the way the reference's
run.py:320 does with ``util.dynamic_load_module(models, prefix="models")``."""

import importlib
import pkgutil


def load_all():
    for _, modname, _ in pkgutil.iter_modules(__path__):
        if not modname.startswith("_") and modname != "model":
            importlib.import_module(__name__ + "." + modname)
