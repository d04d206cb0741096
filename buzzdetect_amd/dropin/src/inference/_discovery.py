"""Shared plugin discovery for embedders and models.

Contract (reference src/inference/embedding.py:40-79 and src/inference/models.py:40-79):
``<root>/<name>/<filename>`` is executed as a module named ``<name>_<suffix>``; the plugin class
is the first attribute, in ``dir()`` (alphabetical) order, that is a strict subclass of the
given base; it is built with ``framehop_prop=...`` and optionally initialised.
"""
import importlib.util
from pathlib import Path


def first_subclass(module, base):
    for attr in dir(module):
        candidate = getattr(module, attr)
        if isinstance(candidate, type) and candidate is not base and issubclass(candidate, base):
            return candidate
    return None


def load_plugin_module(root, name, filename, suffix):
    location = Path(root) / name / filename
    spec = importlib.util.spec_from_file_location(f"{name}_{suffix}", location)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module
