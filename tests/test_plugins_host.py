"""CPU: the plugin boundary mirrors the reference's registry / config-spec behaviour."""
from types import SimpleNamespace

import pytest

import macvo_amd.plugins as P
from macvo_amd import interfaces as I


def test_plugins_register_by_name_and_validate_reference_yaml_args():
    # args copied from Config/Experiment/MACVO/MACVO_Fast.yaml:40-104 (only `type:` changes)
    kp = SimpleNamespace(type="HIP_CovAwareSelector_NoDepth", args=SimpleNamespace(device="cuda", kernel_size=7, mask_width=32, max_match_cov=100.0))
    I.IKeypointSelector.is_valid_config(kp)
    assert I.IKeypointSelector.get_class("HIP_CovAwareSelector_NoDepth") is P.HIP_CovAwareSelector_NoDepth
    mp = SimpleNamespace(type="HIP_MappingPointSelector", args=SimpleNamespace(max_depth=5.0, max_depth_cov=0.005, mask_width=32))
    I.IKeypointSelector.is_valid_config(mp)
    cov = SimpleNamespace(type="HIP_MatchCovariance", args=SimpleNamespace(device="cuda", kernel_size=31, match_cov_default=0.25, min_depth_cov=0.05, min_flow_cov=0.25))
    I.ICovariance2to3.is_valid_config(cov)
    opt = SimpleNamespace(type="HIP_TwoFrame_PGO", args=SimpleNamespace(device="cpu", vectorize=True, parallel=True, graph_type="disp", autodiff=False))
    I.IOptimizer.is_valid_config(opt)
    full = SimpleNamespace(type="HIP_CovAwareSelector", args=SimpleNamespace(device="cuda", kernel_size=7, mask_width=32, max_depth="auto", max_depth_cov=250.0, max_match_cov=100.0))
    I.IKeypointSelector.is_valid_config(full)


def test_config_spec_exact_key_set_and_predicates():
    ok = SimpleNamespace(device="cuda", kernel_size=7, mask_width=32, max_match_cov=100.0)
    P.HIP_CovAwareSelector_NoDepth.is_valid_config(ok)
    with pytest.raises(KeyError):   # excessive key
        P.HIP_CovAwareSelector_NoDepth.is_valid_config(SimpleNamespace(**vars(ok), extra=1))
    with pytest.raises(KeyError):   # missing key
        P.HIP_CovAwareSelector_NoDepth.is_valid_config(SimpleNamespace(device="cuda", kernel_size=7, mask_width=32))
    with pytest.raises(ValueError):  # even kernel
        P.HIP_CovAwareSelector_NoDepth.is_valid_config(SimpleNamespace(device="cuda", kernel_size=6, mask_width=32, max_match_cov=100.0))
    with pytest.raises(ValueError):
        P.HIP_TwoFrame_PGO.is_valid_config(SimpleNamespace(device="cpu", vectorize=True, parallel=True, graph_type="bundle", autodiff=False))


def test_registry_rules():
    with pytest.raises(KeyError):
        I.IKeypointSelector.get_class("NoSuchSelector")
    with pytest.raises(KeyError):   # registered under its own interface only
        I.ICovariance2to3.get_class("HIP_CovAwareSelector")
    with pytest.raises(NameError):  # duplicate names are rejected
        class HIP_MatchCovariance(I.ICovariance2to3):  # noqa: F811
            def estimate(self, *a):
                return None
    if not I.USING_REFERENCE:
        with pytest.raises(ValueError):
            I.IKeypointSelector.is_valid_config(SimpleNamespace(typ="x"))


def test_retrieve_pixels_contract():
    import torch

    m = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    uv = torch.tensor([[1.9, 2.2], [4.0, 0.0]])
    got = I.IFrontend.retrieve_pixels(uv, m)
    assert got.shape == (3, 2) and torch.equal(got[:, 0], m[0, :, 2, 1]) and torch.equal(got[:, 1], m[0, :, 0, 4])
    assert I.IFrontend.retrieve_pixels(uv, None) is None


def test_frontend_plugin_registers_and_explains_missing_network():
    from types import SimpleNamespace

    from macvo_amd import plugins
    from macvo_amd.interfaces import IFrontend

    assert IFrontend.get_class("HIP_FlowFormerCovFrontend") is plugins.HIP_FlowFormerCovFrontend
    assert IFrontend.get_class("HIP_CUDAGraph_FlowFormerCovFrontend") is plugins.HIP_CUDAGraph_FlowFormerCovFrontend
    good = SimpleNamespace(weight="w.pth", device="cuda", dec_dtype="fp32", enc_dtype="fp16", enforce_positive_disparity=False,
                           decoder_depth=12)
    plugins.HIP_FlowFormerCovFrontend.is_valid_config(good)
    with pytest.raises(Exception):
        plugins.HIP_FlowFormerCovFrontend.is_valid_config(SimpleNamespace(**{**vars(good), "device": "cpu"}))
    with pytest.raises(ImportError, match="S_FlowFormer"):
        plugins.HIP_FlowFormerCovFrontend(good)      # FlowFormer source is an empty submodule in the reference checkout


def test_install_flowformer_hooks_on_plain_stand_ins_and_on_the_host_network():
    """The hook installer walks whatever it is given: a plain object carrying only ``memory_encoder.corr`` (no ``nn.Module`` API — the form
    tests/test_gpu_fastmode.py uses) gets that one method rebound; the FlowFormerCov-shaped host network gets all four, with the cost patch
    embedding found under ``memory_encoder.cost_perceiver_encoder`` (on a CPU model the weight pack is refused and ``proj`` stays as it is).
    Binding is lazy: nothing here touches a GPU."""
    import os
    import sys
    from types import SimpleNamespace as NS

    import torch

    from macvo_amd import plugins

    class Enc:
        cfg = NS(cost_heads_num=1)

        def corr(self, a, b):
            raise AssertionError("not rebound")

    assert plugins.install_flowformer_hooks(NS(memory_encoder=Enc())) == ["memory_encoder.corr"]
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import flowformer_host as fh

    torch.manual_seed(0)
    m = fh.FlowFormerCovHost(fh.demo_cfg(decoder_depth=1)).eval()
    names = plugins.install_flowformer_hooks(m)
    assert names[:3] == ["memory_decoder.encode_flow_token", "memory_decoder.upsample_flow", "memory_encoder.corr"]
    assert isinstance(m.memory_encoder.cost_perceiver_encoder.patch_embed.proj, torch.nn.Sequential)      # CPU weights: not packed, not rebound
