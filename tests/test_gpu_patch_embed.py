"""(f)2 — the fused cost patch embedding (csrc/patch_embed.hip) against torch's F.conv2d chain (oracle/patch_embed.py; the FlowFormer submodule is
absent from the reference checkout: parity unpinned against the MAC-VO fork, pinned to the published layer definition).

Both operand types ("f16": IEEE half, the default for fp16 / fp32 encoders; "bf16": for bf16 encoders).  Two bars each: (1) against the SAME
arithmetic — 16-bit operands, fp32 accumulation (``patch_embed_proj_f16 / _bf16``) — the kernel may differ by accumulation order and by the rare
intermediate value that rounds to the other 16-bit neighbour: 2e-3 (bf16) / 2.5e-4 (f16) of the output scale; (2) against the fp32 chain: the
operand error, 2e-2 (bf16) / 2.5e-3 (f16) of the output scale (the reference's Fast mode runs this encoder in fp16, MACVO_Fast.yaml:73-74; its
fp32 configurations in TF32, Frontend.py:275-277 — fp16's mantissa)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


TWIN_TOL = {"bf16": 2e-3, "f16": 2.5e-4}      # vs the same-arithmetic twin, relative to the output scale
FP32_TOL = {"bf16": 2e-2, "f16": 2.5e-3}      # vs the fp32 chain


def _twin(operand):
    from oracle import patch_embed as ope

    return ope.patch_embed_proj_bf16 if operand == "bf16" else ope.patch_embed_proj_f16


def _volume_slices(S, seed, scale=16.0):
    """slices shaped like rows of a cost volume of N(0,1) features with C = 256: ~N(0, 16^2) with a few large responses"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, 1, 60, 80, generator=g) * scale
    x[:, 0, 7, 9] += 200.0
    return x


@pytest.mark.parametrize("operand", ["f16", "bf16"])
@pytest.mark.parametrize("S,tokens", [(1, False), (2, True), (7, False), (600, True)])
def test_cost_patch_embed_matches_the_conv2d_chain(gpu, S, tokens, operand):
    from macvo_amd import ops
    from oracle import patch_embed as ope

    W = ope.make_weights(seed=S)
    x = _volume_slices(S, seed=S + 1)
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)
    got = ops.cost_patch_embed(x.to(gpu), packed, tokens=tokens).cpu()
    ref_bf = _twin(operand)(x, *W)
    ref_32 = ope.patch_embed_proj(x, *W)
    if tokens:
        ref_bf, ref_32 = ope.to_tokens(ref_bf), ope.to_tokens(ref_32)
    assert got.shape == ref_32.shape == ((S, 80, 64) if tokens else (S, 64, 8, 10))
    scale = ref_32.abs().max().item()
    assert (got - ref_bf).abs().max().item() <= TWIN_TOL[operand] * scale, ((got - ref_bf).abs().max().item(), scale)
    assert (got - ref_32).abs().max().item() <= FP32_TOL[operand] * scale, ((got - ref_32).abs().max().item(), scale)


@pytest.mark.parametrize("operand", ["f16", "bf16"])
def test_cost_patch_embed_layers_one_by_one(gpu, operand):
    """Weights that isolate each layer: identity-like taps make the stack's output a known function of the input, so an indexing error in any
    of the three implicit GEMMs (tap order, stride, halo, channel order) cannot hide behind the others."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 1, 60, 80, generator=g) * 4            # positive: the ReLUs are transparent
    for probe in range(4):
        w1, b1, w2, b2, w3, b3 = [torch.zeros_like(t) for t in ope.make_weights(0)]
        if probe == 0:     # a single tap per layer at distinct (ky, kx), distinct channels
            w1[3, 0, 1, 4] = 1.0
            w2[7, 3, 5, 0] = 1.0
            w3[41, 7, 2, 3] = 1.0
        elif probe == 1:   # every tap of conv1, one channel chain (box filters): halo / padding handling at all four borders
            w1[0, 0] = 1.0 / 36
            w2[0, 0] = 1.0 / 36
            w3[0, 0] = 1.0 / 36
        elif probe == 2:   # channel mixing with random sparse weights + biases
            gg = torch.Generator().manual_seed(9)
            w1 = (torch.rand(16, 1, 6, 6, generator=gg) > 0.7).float() * 0.25
            w2 = (torch.rand(32, 16, 6, 6, generator=gg) > 0.9).float() * 0.125
            w3 = (torch.rand(64, 32, 6, 6, generator=gg) > 0.9).float() * 0.125
            b1, b2, b3 = torch.rand(16, generator=gg), torch.rand(32, generator=gg), torch.rand(64, generator=gg) - 0.5
        else:              # negative pre-activations: the two ReLUs clip, the last layer does not
            w1[2, 0, 0, 0] = -1.0
            b1[2] = 2.0
            w2[5, 2, 3, 3] = 1.0
            b2[5] = -1.0
            w3[9, 5, 1, 1] = -1.0
        W = (w1, b1, w2, b2, w3, b3)
        got = ops.cost_patch_embed(x.to(gpu), ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)).cpu()
        ref = _twin(operand)(x, *W)
        tol = TWIN_TOL[operand] * max(ref.abs().max().item(), 1e-3)
        assert (got - ref).abs().max().item() <= tol, (probe, (got - ref).abs().max().item(), tol)


def test_cost_patch_embed_is_deterministic_and_slice_independent(gpu):
    from macvo_amd import ops
    from oracle import patch_embed as ope

    W = [w.to(gpu) for w in ope.make_weights(3)]
    packed = ops.PatchEmbedWeights(*W)
    x = _volume_slices(515, seed=2).to(gpu)                   # odd count: the last pass carries one live slice
    a = ops.cost_patch_embed(x, packed)
    b = ops.cost_patch_embed(x, packed)
    assert torch.equal(a, b)
    solo = ops.cost_patch_embed(x[301:302].contiguous(), packed)
    assert torch.equal(solo[0], a[301])                        # a slice's tokens do not depend on its neighbour in the pass or on the workgroup
    last = ops.cost_patch_embed(x[514:515].contiguous(), packed)
    assert torch.equal(last[0], a[514])


def test_cost_patch_embed_on_a_real_volume_and_unsupported_sizes(gpu):
    from macvo_amd import ops
    from oracle import corr
    from oracle import patch_embed as ope

    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(1, 256, 60, 80, generator=g), torch.randn(1, 256, 60, 80, generator=g)
    vol = ops.corr_volume(f1.to(gpu), f2.to(gpu))             # [4800, 1, 60, 80]: the kernel's real producer
    W = ope.make_weights(11)
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W])
    assert packed.operand == "f16"                            # fp32 / fp16 weights: IEEE half operands; bf16 weights: bf16
    assert ops.PatchEmbedWeights(*[w.to(gpu).bfloat16() for w in W]).operand == "bf16"
    got = ops.cost_patch_embed(vol, packed, tokens=True)
    idx = torch.tensor([0, 1, 2399, 4798, 4799])
    ref = ope.to_tokens(ope.patch_embed_proj_f16(vol[idx.to(gpu)].cpu(), *W))
    assert (got[idx.to(gpu)].cpu() - ref).abs().max().item() <= TWIN_TOL["f16"] * ref.abs().max().item()
    # 64-row slices (what PatchEmbed.forward hands to `proj` after its F.pad, and real 640x512 frames): rows 60..63 carry DATA here
    x64 = torch.randn(6, 1, 64, 80, generator=g) * 16
    for operand in ("f16", "bf16"):
        got64 = ops.cost_patch_embed(x64.to(gpu), ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)).cpu()
        ref64 = _twin(operand)(x64, *W)
        assert got64.shape == (6, 64, 8, 10) and (got64 - ref64).abs().max().item() <= TWIN_TOL[operand] * ref64.abs().max().item()
    assert ops.cost_patch_embed_supported(90, 160) and ops.cost_patch_embed_supported(80, 80)      # round 5: the strip-mined kernel
    assert not ops.cost_patch_embed_supported(24, 32)
    with pytest.raises(ops.L.MacvoHipError):
        ops.cost_patch_embed(torch.zeros(2, 1, 24, 32, device=gpu), ops.PatchEmbedWeights(*[w.to(gpu) for w in W]))


def test_flowformer_hook_rebinds_the_patch_embed_proj(gpu):
    """install_flowformer_hooks on a model shaped like FlowFormer's MemoryEncoder (patch_embed.proj = the three Conv2d layers): the rebound proj
    returns what the original layers return (bf16 bar), also for the 64-row padded input PatchEmbed.forward hands it."""
    import torch.nn as nn
    import torch.nn.functional as F

    from macvo_amd import plugins

    class PatchEmbed(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Sequential(nn.Conv2d(1, 16, 6, 2, 2), nn.ReLU(), nn.Conv2d(16, 32, 6, 2, 2), nn.ReLU(), nn.Conv2d(32, 64, 6, 2, 2))

        def forward(self, x):
            x = F.pad(x, (0, (8 - x.shape[-1] % 8) % 8, 0, (8 - x.shape[-2] % 8) % 8))
            return self.proj(x)

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.patch_embed = PatchEmbed()

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.memory_encoder = Enc()

    torch.manual_seed(0)
    m = Model().to(gpu).eval()
    keys = list(m.state_dict().keys())
    x = _volume_slices(9, seed=4).to(gpu)
    proj = m.memory_encoder.patch_embed.proj
    layers = lambda t: nn.Sequential.forward(proj, t)      # noqa: E731 - the original layers, whatever proj.forward is bound to
    with torch.no_grad():
        want = m.memory_encoder.patch_embed(x)
        done = plugins.install_flowformer_hooks(m)
        assert "memory_encoder.patch_embed.proj" in done
        assert isinstance(proj, nn.Sequential) and list(m.state_dict().keys()) == keys       # nothing re-parented
        # default policy: fp32 slices keep the fp32 layers (no silent rounding of an fp32 model to 16-bit operands) ...
        assert torch.equal(m.memory_encoder.patch_embed(x), want)
        # ... 16-bit slices go through the fused kernel in their own type, cells in and tokens out
        for dt, operand in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
            m16 = Model().to(gpu).eval()
            m16.load_state_dict(m.state_dict())
            m16 = m16.to(dt)
            want16 = m16.memory_encoder.patch_embed(x.to(dt))
            plugins.install_flowformer_hooks(m16)
            got16 = m16.memory_encoder.patch_embed(x.to(dt))
            assert got16.dtype == dt and got16.shape == want16.shape
            # both sides round the same 16-bit weights / cells; the layers also round every conv output to 16 bits, the kernel only the
            # intermediate maps and the tokens: bar = the operand bar + one output rounding
            bar = (FP32_TOL[operand] + (2 ** -8 if operand == "bf16" else 2 ** -11)) * want16.float().abs().max().item()
            assert (got16.float() - want16.float()).abs().max().item() <= bar, operand
    # opt-in for fp32 models: IEEE-half operands
    torch.manual_seed(0)
    m = Model().to(gpu).eval()
    proj = m.memory_encoder.patch_embed.proj
    with torch.no_grad():
        done = plugins.install_flowformer_hooks(m, fuse_patch_embed=True)
        assert "memory_encoder.patch_embed.proj" in done
        got = m.memory_encoder.patch_embed(x)
        small = m.memory_encoder.patch_embed(torch.randn(2, 1, 24, 32, device=gpu))    # a size the kernel does not cover: original layers
        x64 = torch.randn(5, 1, 64, 80, device=gpu) * 8                                # 640x512 frames: rows 60..63 are data, not padding
        got64 = m.memory_encoder.patch_embed(x64)
        want64 = layers(x64)
        assert got.shape == want.shape == (9, 64, 8, 10) and small.shape == (2, 64, 3, 4)
        assert (got - want).abs().max().item() <= FP32_TOL["f16"] * want.abs().max().item()      # fp32 layers -> IEEE-half operands
        assert not torch.equal(got, want)                                                       # ... and it really was the kernel
        assert (got64 - want64).abs().max().item() <= FP32_TOL["f16"] * want64.abs().max().item()
        # a weight update after hooking (load_state_dict, optimizer step, .to()): the packed fragments follow (they were a stale snapshot)
        sd = {k: v * 0.5 for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        want_half = layers(F.pad(x, (0, 0, 0, 4)))
        got_half = m.memory_encoder.patch_embed(x)
        assert (got_half - want_half).abs().max().item() <= FP32_TOL["f16"] * want_half.abs().max().item()
        assert (got_half - got).abs().max().item() > 0.05 * got.abs().max().item()
        # under autocast the Conv2d stack returns the autocast dtype — so does the fused call on fp32 slices; a sliced view whose storage offset
        # breaks the kernel's 16-byte alignment falls through to the layers instead of raising
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ac = m.memory_encoder.patch_embed(x)
        assert ac.dtype == torch.bfloat16 and (ac.float() - got_half).abs().max().item() <= 2 ** -7 * got_half.abs().max().item()
        buf = torch.zeros(9 * 64 * 80 + 1, device=gpu)
        buf[1:] = F.pad(x, (0, 0, 0, 4)).flatten()
        odd = buf[1:].view(9, 1, 64, 80)
        assert odd.data_ptr() % 16 != 0 and torch.equal(proj(odd), layers(odd))
    # autograd: a call that could need gradients runs the layers (the kernel is inference-only)
    xg = x[:2].clone().requires_grad_(True)
    y = m.memory_encoder.patch_embed(xg)
    assert y.requires_grad and torch.equal(y, layers(F.pad(xg, (0, 0, 0, 4))))


@pytest.mark.parametrize("operand,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("tokens", [False, True])
def test_cost_patch_embed_16bit_cells_in_16bit_tokens_out(gpu, operand, dt, tokens):
    """Row (f)2 as SURVEY words it: the volume stays 16-bit between its producer and this consumer.  (1) 16-bit cells in, fp32 out == the fp32-cell
    kernel on the widened cells, bit for bit (the cells reach the matrix pipe as the same 16-bit values either way).  (2) 16-bit tokens == that fp32
    result rounded once to the token type (RNE), bit for bit — both layouts (the token-major form swaps halves between neighbouring lanes).  (3) and
    the whole thing against the oracle's same-arithmetic twin."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    W = ope.make_weights(seed=21)
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)
    for S, H2 in ((1, 60), (301, 60), (6, 64)):
        g = torch.Generator().manual_seed(S)
        x16 = (torch.randn(S, 1, H2, 80, generator=g) * 16).to(dt).to(gpu)
        f32 = ops.cost_patch_embed(x16.float(), packed, tokens=tokens)
        mixed = ops.cost_patch_embed(x16, packed, tokens=tokens, out_dtype=torch.float32)
        assert mixed.dtype == torch.float32 and torch.equal(mixed, f32)
        t16 = ops.cost_patch_embed(x16, packed, tokens=tokens)
        assert t16.dtype == dt and torch.equal(t16, f32.to(dt)), (S, H2)
        ref = _twin(operand)(x16.float().cpu(), *W)
        ref = ope.to_tokens(ref) if tokens else ref
        assert (f32.cpu() - ref).abs().max().item() <= TWIN_TOL[operand] * ref.abs().max().item()
    with pytest.raises(ops.L.MacvoHipError):      # a 16-bit type that is not the operand type, and fp32 cells -> 16-bit tokens: not built, said loudly
        ops.cost_patch_embed(x16.to(torch.bfloat16 if dt == torch.float16 else torch.float16), packed)
    with pytest.raises(ops.L.MacvoHipError):
        ops.cost_patch_embed(x16.float(), packed, out_dtype=dt)


def test_cost_patch_embed_saturates_instead_of_overflowing_fp16(gpu):
    """an un-normalised 256-channel dot product can exceed fp16's range.  Cells and activations beyond +-65504 saturate (finite
    tokens, equal to the twin on clamped values) instead of turning into inf and, a layer later, NaN."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    W = ope.make_weights(seed=4, scale=1.0)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 1, 60, 80, generator=g) * 16
    x[0, 0, 10:14, 20:30] = 3.0e5          # cells beyond fp16
    x[1, 0, 30, 40] = -1.0e6
    x[2] *= 3000.0                         # a whole slice whose conv activations leave fp16's range
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand="f16")
    got = ops.cost_patch_embed(x.to(gpu), packed).cpu()
    assert torch.isfinite(got).all()
    # the twin with the same saturation at every 16-bit rounding
    sat = lambda t: t.clamp(-65504.0, 65504.0).half().float()      # noqa: E731
    import torch.nn.functional as F
    w1, b1, w2, b2, w3, b3 = W
    r = lambda t: t.half().float()      # noqa: E731
    y = sat(F.pad(x, (0, 0, 0, 4)))
    y = sat(F.relu(F.conv2d(y, r(w1), b1, stride=2, padding=2)))
    y = sat(F.relu(F.conv2d(y, r(w2), b2, stride=2, padding=2)))
    ref = F.conv2d(y, r(w3), b3, stride=2, padding=2)
    assert (got - ref).abs().max().item() <= TWIN_TOL["f16"] * ref.abs().max().item()
    assert got[3].abs().max() < 1e3 and got[2].abs().max() > 1e4      # the ordinary slice is untouched by its neighbours' magnitudes


def _slices(S, H2, W2, seed, scale=16.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, 1, H2, W2, generator=g) * scale
    x[:, 0, H2 // 3, W2 // 5] += 200.0
    x[:, 0, H2 - 1, W2 - 1] -= 150.0          # the last cell: the padded rows / columns behind it must read as zeros
    x[:, 0, 0, 0] += 120.0
    return x


@pytest.mark.parametrize("operand", ["f16", "bf16"])
@pytest.mark.parametrize("H2,W2,S,tokens", [(80, 80, 5, False), (80, 80, 300, True), (90, 160, 3, True), (90, 160, 2, False), (90, 160, 700, True),
                                            (96, 160, 4, True), (60, 80, 9, True), (64, 80, 6, False)])
def test_strip_mined_kernel_any_slice_size(gpu, monkeypatch, operand, H2, W2, S, tokens):
    """the slice sizes the whole-slice plan does not cover — 80 x 80 (the reference's 640 x 640 fixture: 100 tokens, 6.25 tiles) and
    90 x 160 (1280 x 720, BASELINE configs[2]: three strips of four token rows, halo rows recomputed), plus the padded 96 x 160 — through
    patch_embed_v2.hip against the conv2d chain in the same arithmetic and in fp32; and the same kernel forced onto 60 / 64 x 80 (MV_PE_STRIP=1)."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    monkeypatch.setenv("MV_PE_STRIP", "1")
    W = ope.make_weights(seed=H2 + S)
    x = _slices(S, H2, W2, seed=S + 2)
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)
    got = ops.cost_patch_embed(x.to(gpu), packed, tokens=tokens).cpu()
    idx = torch.arange(S) if S <= 16 else torch.tensor([0, 1, S // 3, S // 2, S - 2, S - 1])
    ref_16 = _twin(operand)(x[idx], *W)
    ref_32 = ope.patch_embed_proj(x[idx], *W)
    if tokens:
        ref_16, ref_32 = ope.to_tokens(ref_16), ope.to_tokens(ref_32)
    h, w = (H2 + 7) // 8, (W2 + 7) // 8
    assert got.shape == ((S, h * w, 64) if tokens else (S, 64, h, w))
    scale = ref_32.abs().max().item()
    assert (got[idx] - ref_16).abs().max().item() <= TWIN_TOL[operand] * scale, ((got[idx] - ref_16).abs().max().item(), scale)
    assert (got[idx] - ref_32).abs().max().item() <= FP32_TOL[operand] * scale
    # deterministic; a slice's tokens do not depend on its neighbours or on which workgroup / strip order produced them
    assert torch.equal(ops.cost_patch_embed(x.to(gpu), packed, tokens=tokens).cpu(), got)
    solo = ops.cost_patch_embed(x[S // 2: S // 2 + 1].to(gpu), packed, tokens=tokens).cpu()
    assert torch.equal(solo[0], got[S // 2])


@pytest.mark.parametrize("H2,W2", [(80, 80), (90, 160)])
def test_strip_mined_kernel_16bit_cells_and_tokens(gpu, H2, W2):
    """the Fast-mode form of the new sizes: fp16 cells in / fp16 tokens out == the fp32-cell form on the widened cells, rounded once (both layouts)."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    W = ope.make_weights(seed=3)
    for operand, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)
        x16 = _slices(7, H2, W2, seed=11).to(dt).to(gpu)
        for tokens in (False, True):
            f32 = ops.cost_patch_embed(x16.float(), packed, tokens=tokens)
            assert torch.equal(ops.cost_patch_embed(x16, packed, tokens=tokens, out_dtype=torch.float32), f32)
            t16 = ops.cost_patch_embed(x16, packed, tokens=tokens)
            assert t16.dtype == dt and torch.equal(t16, f32.to(dt))
        ref = _twin(operand)(x16.float().cpu(), *W)
        assert (ops.cost_patch_embed(x16.float(), packed).cpu() - ref).abs().max().item() <= TWIN_TOL[operand] * ref.abs().max().item()


def test_strip_mined_layers_one_by_one_at_720p(gpu):
    """single taps per layer at distinct (ky, kx) / channels on a 90 x 160 slice: an indexing error in any of the three implicit GEMMs, in the strip
    windows' row origins or in the recomputed halo rows cannot hide (positive inputs: the ReLUs are transparent)."""
    from macvo_amd import ops
    from oracle import patch_embed as ope

    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 90, 160, generator=g) * 4
    for probe in range(3):
        w1, b1, w2, b2, w3, b3 = [torch.zeros_like(t) for t in ope.make_weights(0)]
        if probe == 0:
            w1[3, 0, 1, 4] = 1.0; w2[7, 3, 5, 0] = 1.0; w3[41, 7, 2, 3] = 1.0
        elif probe == 1:
            w1[0, 0] = 1.0 / 36; w2[0, 0] = 1.0 / 36; w3[0, 0] = 1.0 / 36
        else:
            gg = torch.Generator().manual_seed(9)
            w1 = (torch.rand(16, 1, 6, 6, generator=gg) > 0.7).float() * 0.25
            w2 = (torch.rand(32, 16, 6, 6, generator=gg) > 0.9).float() * 0.125
            w3 = (torch.rand(64, 32, 6, 6, generator=gg) > 0.9).float() * 0.125
            b1, b2, b3 = torch.rand(16, generator=gg), torch.rand(32, generator=gg), torch.rand(64, generator=gg) - 0.5
        W = (w1, b1, w2, b2, w3, b3)
        for tokens in (False, True):
            got = ops.cost_patch_embed(x.to(gpu), ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand="f16"), tokens=tokens).cpu()
            ref = ope.patch_embed_proj_f16(x, *W)
            ref = ope.to_tokens(ref) if tokens else ref
            tol = TWIN_TOL["f16"] * max(ref.abs().max().item(), 1e-3)
            assert (got - ref).abs().max().item() <= tol, (probe, tokens, (got - ref).abs().max().item(), tol)


@pytest.mark.parametrize("operand,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("H2", [60, 64])
def test_pipelined_kernel_equals_the_phase_kernel(gpu, monkeypatch, operand, dt, H2):
    """patch_embed_v3.hip (two wave groups: conv1 + conv2 of slice k beside conv3 of slice k - 1, conv2 maps as the ping-pong) against the phase-by-phase kernel
    of patch_embed.hip (round 6: a TEST-ONLY build, -DMV_PE_PHASE_REFERENCE, tests/pe_phase_ref.py — the product library no longer carries it) on the same slices: the same 16-bit values meet in the same k order per output, so the tokens are BIT-identical — fp32 and 16-bit cells, both token
    types, both layouts, slice counts around the persistent grid (1, odd, 256 + tail)."""
    from macvo_amd import ops
    from oracle import patch_embed as ope
    from tests import pe_phase_ref

    W = ope.make_weights(seed=31)
    packed = ops.PatchEmbedWeights(*[w.to(gpu) for w in W], operand=operand)
    for S in (1, 7, 300, 515):
        x = _slices(S, H2, 80, seed=S).to(gpu)
        for tokens in (False, True):
            for xin, odt in ((x, None), (x.to(dt), None), (x.to(dt), torch.float32)):
                a = pe_phase_ref.cost_patch_embed(xin, packed, tokens=tokens, out_dtype=odt)     # the phase kernel: test-only build of patch_embed.hip
                b = ops.cost_patch_embed(xin, packed, tokens=tokens, out_dtype=odt)
                assert a.dtype == b.dtype and torch.equal(a, b), (S, tokens, xin.dtype, odt)
    ref = _twin(operand)(x[:4].cpu(), *W)
    got = ops.cost_patch_embed(x[:4].contiguous(), packed).cpu()
    assert (got - ref).abs().max().item() <= TWIN_TOL[operand] * ref.abs().max().item()
