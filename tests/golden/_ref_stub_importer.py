"""Stub importer used ONLY by tests/golden/make_golden.py (fixture generation, run once in
the build container where /root/reference is mounted).

The reference package `neusky` imports nerfstudio / reni / nerfacc / tinycudann / ... at module
import time; none of them is installed here.  This finder fabricates permissive
placeholder modules for those top-level packages so that the reference's *in-tree* pure-torch
functions can be imported and called unbound.  Nothing from the reference is copied: this file
only teaches `import` how to not fail.  It is never imported by the product, the tests or the bench.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import sys
import types

MISSING = {
    "nerfstudio", "reni", "nerfacc", "tinycudann", "cv2", "jaxtyping", "icosphere",
    "torchvision", "torchmetrics", "tyro", "torchtyping", "rich", "wandb", "imageio",
    "matplotlib", "mediapy", "pyexr", "OpenEXR", "Imath", "skimage", "plotly", "viser",
}


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = _make_placeholder(f"{cls.__name__}.{name}")
        type.__setattr__(cls, name, obj)  # stable identity: usable as a dict key (e.g. FieldHeadNames.RGB)
        return obj

    def __getitem__(cls, item):
        return cls

    def __call__(cls, *a, **k):
        # behave like a decorator when handed exactly one function/class (e.g. profiler.time_function)
        if len(a) == 1 and not k and (isinstance(a[0], (types.FunctionType, type))) and cls.__name__.endswith(
            ("time_function", "check_main_thread", "decorate_all")
        ):
            return a[0]
        return super().__call__(*a, **k)

    def __iter__(cls):
        return iter(())


def _make_placeholder(name: str):
    def __init__(self, *a, **k):
        self._args = a
        self._kwargs = k
        for kk, vv in k.items():
            try:
                object.__setattr__(self, kk, vv)
            except Exception:
                pass

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _make_placeholder(f"{name}.{item}")()

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and isinstance(a[0], (types.FunctionType, type)):
            return a[0]
        return _make_placeholder(f"{name}()")()

    def __getitem__(self, item):
        return self

    def __iter__(self):
        return iter(())

    return _Meta(
        name.split(".")[-1] or "Placeholder",
        (object,),
        {
            "__init__": __init__,
            "__getattr__": __getattr__,
            "__call__": __call__,
            "__getitem__": __getitem__,
            "__iter__": __iter__,
            "__module__": "_refstub",
        },
    )


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = _make_placeholder(f"{self.__name__}.{name}")
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        m.__all__ = []
        return m

    def exec_module(self, module):
        return None


def install(reference_root: str = "/root/reference") -> None:
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference mount
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
