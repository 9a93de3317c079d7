"""Overlay of this package on a reference checkout (the drop-in boundary, SURVEY.md §8b / INTEGRATION.md).

The reference's entry points (`inference_main.py`, `train.py`, `flask_api.py`, `webUI.py`) import the hot path by MODULE
NAME: `models`, `utils`, `modules.*`, `vdecoder.*`, `vencoder.*`, `inference.infer_tool`, `diffusion.*`, `data_utils`.
With this package's directory AHEAD of the reference checkout on `sys.path` those names resolve to the MI355X engine; the
names the engine does not mirror (`cluster`, `spkmix`, `modules.F0Predictor.*`, `modules.enhancer`, `inference.slicer` when
librosa is wanted, the ONNX / whisper / wavlm encoders ...) must keep resolving to the reference's own files.  Top-level
modules do that by themselves (they are simply found further down `sys.path`); SUBMODULES of the shadowed packages do not,
because Python binds a package to the first directory that provides it.  `OverlayFinder` closes that gap: it sits at the
END of `sys.meta_path`, so it is consulted only after the normal import of e.g. `modules.F0Predictor` failed inside this
package, and then looks for `<entry>/modules/F0Predictor(.py|/__init__.py)` under the other `sys.path` entries.

`python so-vits-svc_amd/svc_run.py <reference script> [args...]` (svc_run.py) is the launcher that sets the path order up:
a script run as `python inference_main.py` gets ITS OWN directory as `sys.path[0]`, ahead of PYTHONPATH, so PYTHONPATH
alone cannot put the engine first.
"""
import importlib.abc
import importlib.util
import os
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
OVERLAID = ("modules", "vdecoder", "vencoder", "inference", "diffusion")


def reference_roots():
    """sys.path entries (other than this package) that look like a so-vits-svc checkout."""
    out = []
    for e in sys.path:
        d = os.path.abspath(e or os.getcwd())
        if d == PKG_DIR or d in out:
            continue
        if os.path.isfile(os.path.join(d, "models.py")) and os.path.isdir(os.path.join(d, "vdecoder")):
            out.append(d)
    return out


class OverlayFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        parts = fullname.split(".")
        if len(parts) < 2 or parts[0] not in OVERLAID:
            return None
        rel = os.path.join(*parts)
        for root in reference_roots():
            pkg_init = os.path.join(root, rel, "__init__.py")
            if os.path.isfile(pkg_init):
                return importlib.util.spec_from_file_location(fullname, pkg_init,
                                                              submodule_search_locations=[os.path.join(root, rel)])
            if os.path.isdir(os.path.join(root, rel)):          # namespace-style directory without __init__.py
                spec = importlib.util.spec_from_loader(fullname, loader=None, is_package=True)
                spec.submodule_search_locations = [os.path.join(root, rel)]
                return spec
            mod = os.path.join(root, rel + ".py")
            if os.path.isfile(mod):
                return importlib.util.spec_from_file_location(fullname, mod)
        return None


_installed = None


def install():
    """Idempotent: append the finder to sys.meta_path (after the standard finders)."""
    global _installed
    if _installed is None or _installed not in sys.meta_path:
        _installed = OverlayFinder()
        sys.meta_path.append(_installed)
    return _installed


def load_reference_module(name, alias):
    """Load the reference checkout's top-level `<name>.py` (shadowed by this package's module of the same name) under the
    module name `alias`; None when no checkout is on sys.path."""
    if alias in sys.modules:
        return sys.modules[alias]
    for root in reference_roots():
        f = os.path.join(root, name + ".py")
        if os.path.isfile(f):
            spec = importlib.util.spec_from_file_location(alias, f)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[alias] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(alias, None)
                raise
            return mod
    return None
