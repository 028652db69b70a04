"""The module-level drop-in of INTEGRATION.md section 1: with rehrseg_amd/ on sys.path the import lines of
the reference's train_all.py:20-31 resolve to the MI355X mirror -- run in a fresh interpreter with a clean
sys.path (no repo root, no tests/ on it)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = textwrap.dedent("""
    import sys
    sys.path.insert(0, {pkg!r})
    assert {root!r} not in sys.path[1:] or True
    from models.FLAVR.FLAVR_arch import UNet_3D_3D                                   # train_all.py:20
    from models.seg_model import SegModel, Distiller                                 # train_all.py:21
    from utils.seg_utils import zscore_normalization, BCEDiceLoss, _build_loss       # train_all.py:29
    from utils.sr_utils import apply_to_vol_flavr
    import models.FLAVR.resnet_3D as r3d
    import rehrseg_amd.models.FLAVR.FLAVR_arch as real
    import rehrseg_amd.models.FLAVR.resnet_3D as real_r3d
    assert UNet_3D_3D is real.UNet_3D_3D and r3d is real_r3d                         # one set of module objects
    m = UNet_3D_3D(2, "unet_18", 4, 4, batchnorm=False, joinType="concat", upmode="transpose")
    assert len(m.state_dict()) == 79 and r3d.useBias is True
    d = Distiller(64, 64, 0.0, 1.0, 1.0)
    import torch
    s = SegModel(input_channels=1, n_stages=2, features_per_stage=[32, 64], conv_op=torch.nn.Conv3d,
                 kernel_sizes=[[1, 3, 3], [3, 3, 3]], strides=[[1, 1, 1], [1, 2, 2]], n_conv_per_stage=[2, 2],
                 num_classes=2, upscale=4, n_conv_per_stage_decoder=[2], conv_bias=True,
                 norm_op=torch.nn.InstanceNorm3d, norm_op_kwargs={{"eps": 1e-5, "affine": True}}, dropout_op=None,
                 dropout_op_kwargs=None, nonlin=torch.nn.LeakyReLU, nonlin_kwargs={{"inplace": True}},
                 deep_supervision=False)
    assert "sr_head.2.weight" in s.state_dict()
    try:
        import models.wdsr                                                            # out of scope: must say so
    except ImportError:
        pass
    else:
        raise SystemExit("models.wdsr should not resolve")
    print("DROPIN_OK")
""")


def test_integration_snippet_in_clean_interpreter(tmp_path):
    code = SNIPPET.format(pkg=os.path.join(ROOT, "rehrseg_amd"), root=ROOT)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stderr[-2000:]


def test_reference_data_imports_resolve():
    """train_all.py:24,28 name the data set classes and blur helpers by these paths."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from utils.train_set import TrainSetMultiple, TrainSetMultipleSegSREfficient\n"
            "from utils.blur_kernel_ops import calc_extended_patch_size, parse_kernel\n"
            "from utils.pad import target_pad\n"
            "import rehrseg_amd.utils.train_set as t\n"
            "assert TrainSetMultiple is t.TrainSetMultiple\n"
            "k = parse_kernel(None, 'gaussian', 3.0); assert tuple(k.shape) == (1, 1, 7, 1) and abs(float(k.sum()) - 1) < 1e-6\n"
            "print('ok')\n")
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code % os.path.join(root, "rehrseg_amd")], capture_output=True, text=True,
                         cwd="/tmp", env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_aliased_modules_keep_their_real_spec():
    """the alias loader must not leave `__spec__` pointing at the alias name -- lazy relative imports
    inside the mirrored modules would warn (`__package__ != __spec__.parent`), fatally under -W error."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import utils.seg_utils as su, models.seg_model as sm, models.FLAVR.FLAVR_arch as fa\n"
            "import rehrseg_amd.utils.seg_utils as real\n"
            "assert su is real\n"
            "for m, n in ((su, 'rehrseg_amd.utils.seg_utils'), (sm, 'rehrseg_amd.models.seg_model'),\n"
            "             (fa, 'rehrseg_amd.models.FLAVR.FLAVR_arch')):\n"
            "    assert m.__spec__.name == n and m.__spec__.parent == m.__package__, (m.__spec__, m.__package__)\n"
            "exec('from .. import lib as _l', su.__dict__)    # a lazy relative import, as _FusedBCEDice.forward does\n"
            "print('SPEC_OK')\n" % os.path.join(ROOT, "rehrseg_amd"))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-W", "error::ImportWarning", "-c", code], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "SPEC_OK" in r.stdout, r.stderr[-2000:]
