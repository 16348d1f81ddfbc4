"""Run one of the reference's scripts UNCHANGED on top of contextflow_amd's `layers` package.

    python -m contextflow_amd.run /path/to/contextflow/contextflow/model.py --dataset cifar10 --coupling conv ...

Python resolves `from layers import *` (contextflow/model.py:14-15) through sys.path, whose first entry is the
script's own directory - where the reference's `layers` lives - so PYTHONPATH cannot rebind it.  This launcher
builds the path the other way round, [<dropin dir holding only `layers`>, <script dir>, ...], and then executes the
script in THIS process with runpy (no exec of another program; nothing here touches the GPU before the script does).
`load()` imports a script as a module instead of running it as __main__ (used to reach `create_model` from tests and
notebooks)."""
import importlib.util
import os
import runpy
import sys

DROPIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def bind(script):
    """Put the drop-in `layers` ahead of everything and the script's directory right behind it."""
    script = os.path.abspath(script)
    if not os.path.isfile(script):
        raise FileNotFoundError(script)
    sdir = os.path.dirname(script)
    loaded = sys.modules.get("layers")
    if loaded is not None and not getattr(loaded, "__file__", "").startswith(DROPIN + os.sep):
        raise RuntimeError("a different `layers` package is already imported (%s); start from a fresh interpreter"
                           % getattr(loaded, "__file__", "?"))
    # '' / cwd entries would shadow the drop-in when the working directory is the reference's own source directory
    sys.path[:] = [DROPIN, sdir] + [q for q in sys.path if q not in (DROPIN, sdir)]
    import layers                                    # bind now: a later sys.path edit by the script cannot rebind it
    assert layers.__file__.startswith(DROPIN + os.sep), layers.__file__
    return script


def load(script, name=None):
    """Import `script` as module `name` (default: its file name) with the drop-in `layers` bound; returns the module."""
    script = bind(script)
    name = name or os.path.splitext(os.path.basename(script))[0]
    spec = importlib.util.spec_from_file_location(name, script)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(name, None)
        raise
    return mod


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0 if argv else 2
    script = bind(argv[0])
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
