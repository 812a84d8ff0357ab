"""The drop-in claim, checked against the reference's OWN source where it is available (the build container; the GPU boxes have
no /root/reference, so these tests skip there): after `dropin.install()` the reference's `mamba_simple.py` files import
unmodified -- every `mamba_ssm` / `causal_conv1d` name they ask for resolves to this package -- and the module they define has
the same parameter names and shapes as the mirror, i.e. checkpoints move both ways.  Construction only: CPU."""
import importlib.util
import os
import sys

import pytest

REF = "/root/reference/CXPMRG_Bench_MambaXray_VL"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")


@pytest.fixture(autouse=True)
def _restore_sys_modules():
    """The drop-in and the timm stand-ins register modules under third-party names; later tests (transformers probes
    `find_spec("timm")`) must not see them."""
    before = dict(sys.modules)
    yield
    for k in list(sys.modules):
        if k not in before:
            del sys.modules[k]
    sys.modules.update({k: v for k, v in before.items() if sys.modules.get(k) is not v})


def _load_reference(path, name):
    import medical_image_analysis_amd.dropin as dropin
    dropin.install()
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _layout(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


@pytest.mark.parametrize("bimamba_type", ["v2", "v3", "v4"])
def test_reference_finetuning_mamba_imports_and_matches_layout(bimamba_type):
    from medical_image_analysis_amd.mamba_simple import Mamba
    ref = _load_reference(os.path.join(REF, "arm/Finetuning/mamba_simple.py"), "ref_ft_mamba_simple")
    for name in ("selective_scan_fn", "mamba_inner_fn", "mamba_inner_fn_no_out_proj", "causal_conv1d_fn", "causal_conv1d_update",
                 "selective_state_update"):
        assert getattr(ref, name).__module__.startswith("medical_image_analysis_amd."), name
    theirs = ref.Mamba(d_model=48, bimamba_type=bimamba_type)
    mine = Mamba(d_model=48, bimamba_type=bimamba_type)
    assert _layout(theirs) == _layout(mine)
    mine.load_state_dict(theirs.state_dict(), strict=True)        # a reference checkpoint loads
    theirs.load_state_dict(mine.state_dict(), strict=True)        # and the other way round


def test_reference_pretrain_mamba_imports_and_matches_layout():
    from medical_image_analysis_amd.mamba_simple import Mamba
    ref = _load_reference(os.path.join(REF, "pretrain/mamba_simple.py"), "ref_pt_mamba_simple")
    theirs = ref.Mamba(d_model=64, expand=1, bimamba_type="None")
    mine = Mamba(d_model=64, expand=1, bimamba_type="None")
    assert _layout(theirs) == _layout(mine)


def _timm_stand_in():
    """timm is not installed in the build container: the few names the reference's model files import from it (DropPath,
    initialisers, registry decorator) come from the golden generator's stand-ins; they do not touch parameter layouts."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    try:
        import make_golden
    finally:
        sys.path.pop(0)
    make_golden.install_timm_stubs()


def test_reference_arm_encoder_layout_equals_mirror():
    """The reference's `arm_base_pz16` (arm/Finetuning/models_mamba.py:398-410) built on top of the drop-in ops next to the
    mirror's: identical state_dict keys and shapes, strict load both ways."""
    from medical_image_analysis_amd import models_mamba as mine_mod
    _timm_stand_in()
    ft = os.path.join(REF, "arm/Finetuning")
    sys.path.insert(0, ft)
    saved = {k: sys.modules.pop(k, None) for k in ("mamba_simple", "rope")}
    try:
        ref = _load_reference(os.path.join(ft, "models_mamba.py"), "ref_ft_models_mamba")
        theirs = ref.arm_base_pz16("base")
    finally:
        sys.path.remove(ft)
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    mine = mine_mod.arm_base_pz16("base")
    assert _layout(theirs) == _layout(mine)
    mine.load_state_dict(theirs.state_dict(), strict=True)
    assert mine.num_features == theirs.num_features == 768 and mine.patch_embed.num_patches == theirs.patch_embed.num_patches == 196


def test_reference_pretrain_model_layout_equals_mirror():
    """`pretrain/models_pretrain.py` arm_base_pz16 (stage-1 VisionMamba, :518-527) likewise."""
    from medical_image_analysis_amd import models_pretrain as mine_mod
    _timm_stand_in()
    pt = os.path.join(REF, "pretrain")
    sys.path.insert(0, pt)
    saved = {k: sys.modules.pop(k, None) for k in ("mamba_simple", "rope", "utils", "utils.pos_embed")}
    try:
        ref = _load_reference(os.path.join(pt, "models_pretrain.py"), "ref_pt_models_pretrain")
        theirs = ref.arm_base_pz16()
    finally:
        sys.path.remove(pt)
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    mine = mine_mod.arm_base_pz16()
    assert _layout(theirs) == _layout(mine)
    mine.load_state_dict(theirs.state_dict(), strict=True)
