"""The drop-in recipe of INTEGRATION.md section 2, driven as documented (CPU, no GPU):

    python -m contextflow_amd.run /path/to/contextflow/contextflow/model.py ...

* the launcher binds `layers` to contextflow_amd's package even when the script's own directory (and the working
  directory) hold another `layers` package - the situation in the reference tree (contextflow/model.py:14-15);
* nothing else of contextflow_amd becomes a top-level name;
* the reference's UNMODIFIED `create_model` (model.py:95-163), imported through the launcher, builds the mnist /
  cifar10 / smap generalists and a specialist out of this repo's classes with the reference's state_dict layout.
  That half needs /root/reference (build container only; skipped elsewhere - the reference never travels)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SCRIPT = "/root/reference/contextflow/model.py"


def _run(args, cwd, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    return subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_binds_layers_ahead_of_the_script_directory(tmp_path):
    """A script directory with its own `layers`, `model`, `build`, `dist` (decoys): the script must see OUR layers and
    ITS OWN everything else."""
    (tmp_path / "layers").mkdir()
    (tmp_path / "layers" / "__init__.py").write_text("DECOY = True\n")
    for name in ("model", "build", "dist"):
        (tmp_path / (name + ".py")).write_text("DECOY = True\n")
    (tmp_path / "script.py").write_text(textwrap.dedent("""
        import sys
        from layers.rtdl.nn._embeddings import *
        from layers import *
        import layers, layers.coupling, layers.autograd_layers, contextflow_amd.layers.coupling as real
        import model, build, dist
        assert model.DECOY and build.DECOY and dist.DECOY            # the script's own modules, not contextflow_amd's
        assert not hasattr(layers, "DECOY"), layers.__file__
        assert layers.coupling is real and Coupling is real.Coupling
        assert layers.autograd_layers is sys.modules["contextflow_amd.layers.autograd_layers"]   # imported on demand
        assert Coupling.__module__ == "contextflow_amd.layers.coupling", Coupling.__module__
        assert OneHotEncoder.__module__ == EyeEncoder.__module__ == CatEmbeddings.__module__ == "contextflow_amd.layers.context"
        assert __name__ == "__main__" and sys.argv[1:] == ["--flag", "7"], sys.argv
        try:
            import layers.no_such_module
        except ModuleNotFoundError:
            pass
        else:
            raise AssertionError("layers.no_such_module imported")
        print("BOUND", layers.__file__)
    """))
    r = _run(["-m", "contextflow_amd.run", str(tmp_path / "script.py"), "--flag", "7"], cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BOUND " + os.path.join(ROOT, "contextflow_amd", "dropin", "layers") in r.stdout, r.stdout


def test_dropin_directory_holds_only_layers():
    d = os.path.join(ROOT, "contextflow_amd", "dropin")
    assert sorted(n for n in os.listdir(d) if not n.startswith("__pycache__")) == ["layers"]


def test_launcher_refuses_a_foreign_layers(tmp_path):
    (tmp_path / "layers").mkdir()
    (tmp_path / "layers" / "__init__.py").write_text("")
    (tmp_path / "s.py").write_text("")
    code = "import sys; sys.path.insert(0, %r); import layers; from contextflow_amd import run; run.load(%r)" % (
        str(tmp_path), str(tmp_path / "s.py"))
    r = _run(["-c", code], cwd=str(tmp_path))
    assert r.returncode != 0 and "different `layers` package" in r.stderr, r.stderr


DRIVER = r"""
import os, sys, types
import torch.nn as nn
sys.dont_write_bytecode = True                      # the reference tree is read-only

def stub(name, **attrs):                            # third-party / dataset modules absent from this image; none is on
    m = types.ModuleType(name); m.__path__ = []     # the density path (SURVEY.md Appendix C)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m

class InputDropout(nn.Module):
    def __init__(self, *a, **k): super().__init__()
    def forward(self, x): return x

stub("datasets"); stub("datasets.ts", InputDropout=InputDropout, load_data_ts=None)
stub("datasets.mnist", load_data=None); stub("datasets.cifar10", load_data=None)
stub("datasets.mtad_dataloader", mtad_entities={})
stub("torchinfo", summary=None); stub("ood_metrics", auroc=None, aupr=None, fpr_at_95_tpr=None)

from contextflow_amd import run
M = run.load(sys.argv[1])                           # the reference's model.py, unmodified, as module `model`
import layers, contextflow_amd as cfa
assert layers.__file__.startswith(run.DROPIN), layers.__file__
assert M.__file__ == sys.argv[1]
assert M.Coupling is cfa.layers.Coupling and M.FlowSequential is cfa.layers.FlowSequential
assert M.OneHotEncoder is cfa.layers.OneHotEncoder and M.CatEmbeddings is cfa.layers.CatEmbeddings
from layers.rtdl.nn._embeddings import *            # model.py:14
from oracle import flow_oracle as fo, params as op

def ref_build(name, generalist=True, contexts=None, enc_emb="onehot", enc_type="uniform", contextflow=False):
    preset, data_size, mixtures = cfa.preset_config(name)
    M.c = types.SimpleNamespace(dataset=name)       # create_model reads this module global (model.py:113,125)
    cfg = dict(preset, generalist=generalist, enc_emb=enc_emb, enc_type=enc_type, contextflow=contextflow)
    return M.create_model(cfg, data_size=data_size, mixtures=mixtures, contexts=contexts), cfg, data_size, mixtures

for name, contexts in (("mnist", [-1]), ("cifar10", [-1, -1]), ("smap", [55])):
    flow, cfg, data_size, mixtures = ref_build(name, contexts=contexts)
    assert type(flow).__module__ == "contextflow_amd.layers.flowsequential", type(flow).__module__
    for m in flow.modules():
        assert type(m).__module__.startswith(("contextflow_amd.", "torch.")), type(m)
    ops, prior, M2 = fo.program(name)
    spec = op.param_spec(ops, prior, M2)            # == the reference's own state_dict (asserted in make_golden.py)
    sd = flow.state_dict()
    assert list(sd) == list(spec), set(sd) ^ set(spec)
    assert all(tuple(sd[k].shape) == tuple(spec[k][0]) for k in sd)
    print("generalist", name, len(sd), sum(v.numel() for v in flow.parameters()))

for name, contexts, emb, enc, cf in (("cifar10", [15, 5], "onehot", "uniform", True), ("smap", [55], "eye", "uniform", False),
                                     ("mnist", [64], "embed", "probsample", True)):
    flow, cfg, data_size, mixtures = ref_build(name, False, contexts, emb, enc, cf)
    assert type(flow).__module__ == "contextflow_amd.layers.flowsequential"
    # model.ContextEncoder (model.py:30-90) is the script's own nn.Sequential; everything inside it is ours
    foreign = {type(m).__qualname__ for m in flow.modules() if not type(m).__module__.startswith(("contextflow_amd.", "torch."))}
    assert foreign == {"ContextEncoder"}, foreign
    ours = cfa.create_model(cfg, data_size, mixtures, contexts)
    a, b = flow.state_dict(), ours.state_dict()
    assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a), set(a) ^ set(b)
    print("specialist", name, emb, enc, cf, len(a))
print("DROPIN OK")
"""


@pytest.mark.skipif(not os.path.isfile(REF_SCRIPT), reason="needs the reference tree (build container only)")
def test_reference_create_model_builds_on_the_dropin():
    """cwd = the reference's own source directory, the hardest case: its `layers`, `model`, `config` ... are all importable
    from '' and from the script directory."""
    r = _run(["-c", DRIVER, REF_SCRIPT], cwd=os.path.dirname(REF_SCRIPT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DROPIN OK" in r.stdout
    assert r.stdout.count("generalist") == 3 and r.stdout.count("specialist") == 3, r.stdout


@pytest.mark.skipif(not os.path.isfile(REF_SCRIPT), reason="needs the reference tree (build container only)")
def test_documented_command_reaches_the_reference_script():
    """`python -m contextflow_amd.run .../model.py --help` as INTEGRATION.md writes it.  In this image the script stops at
    its first import of a package that is not installed (torchvision / skimage / torchinfo ...): that must be the ONLY
    way it fails here - i.e. it got past `from layers import *` on our package - or it prints its argparse help."""
    r = _run(["-m", "contextflow_amd.run", REF_SCRIPT, "--help"], cwd=os.path.dirname(REF_SCRIPT))
    if r.returncode == 0:
        assert "--dataset" in r.stdout
    else:
        assert "ModuleNotFoundError" in r.stderr and "contextflow_amd" not in r.stderr.split("ModuleNotFoundError")[-1], r.stderr[-2000:]
