"""Process-local import recipe for the *reference* python tree (generation-time only).

TEST INFRASTRUCTURE -- used only by tests/golden/make_golden.py in the build container,
where /root/reference exists.  Nothing here is imported by the product, by the -m gpu
tests, by smoke() or by bench.py: the reference cannot travel to the GPU box, only the
.npz vectors this tooling produces do.

The recipe follows SURVEY.md section 8(c)-1: skip sglang/__init__.py, stub the six absent
third-party modules and one internal predicate, never write into /root/reference.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = "/root/reference"

_ABSENT = ("orjson", "pybase64", "zmq", "sgl_kernel", "torch_memory_saver")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


def install():
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    if "sglang" in sys.modules:
        return
    pkg = types.ModuleType("sglang")
    pkg.__path__ = [os.path.join(REF_ROOT, "python", "sglang")]
    sys.modules["sglang"] = pkg
    sys.meta_path.insert(0, _StubFinder())

    msgspec = types.ModuleType("msgspec")

    class Struct:
        def __init_subclass__(cls, **kw):
            super().__init_subclass__()

        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    msgspec.Struct = Struct
    sys.modules["msgspec"] = msgspec

    cgr = types.ModuleType("sglang.srt.model_executor.cuda_graph_runner")
    cgr.get_is_capture_mode = lambda: False
    sys.modules["sglang.srt.model_executor.cuda_graph_runner"] = cgr
