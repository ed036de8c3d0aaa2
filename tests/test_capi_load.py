"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/ganet_hip.h declares; host-side validation paths need no GPU."""
import ctypes
import os
import re

import pytest

from ganet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ganet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ganet_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_one_hip_runtime_in_the_process():
    """The binding loaded BEFORE torch must still end up on torch's HIP runtime: with two
    libamdhip64 in one process every launch on a torch stream fails (GANET_ERR_HIP), which
    is what build() followed by smoke() in one process used to do."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ganet_amd import _lib\n"
        "_lib.lib()\n"
        "import torch\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print(len(libs), libs)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == "1", out.stdout


def test_status_strings_and_version():
    L = _lib.lib()
    assert L.ganet_abi_version() == 1
    assert _lib.status_string(0) == "ok"
    for code in range(-7, 0):
        assert _lib.status_string(code) != "unknown status"
    assert _lib.status_string(-99) == "unknown status"


def test_out_size_rule():
    L = _lib.lib()
    # (in + 2*pad - (dil*(k-1)+1)) // stride + 1   (deform_conv.py:98-111)
    assert L.ganet_dcn_out_size(100, 3, 1, 7, 1) == 100
    assert L.ganet_dcn_out_size(40, 0, 1, 1, 1) == 40
    assert L.ganet_dcn_out_size(7, 1, 2, 3, 2) == 3
    assert L.ganet_dcn_out_size(2, 0, 1, 7, 1) == 0


def _dcn_params(**kw):
    p = _lib.DcnParams()
    base = dict(B=2, Cin=64, H=8, W=16, Cout=64, kH=1, kW=7, sH=1, sW=1, pH=0,
                pW=3, dH=1, dW=1, groups=1, deformable_groups=1, dtype=0,
                x_layout=0, out_layout=0)
    base.update(kw)
    for k, v in base.items():
        setattr(p, k, v)
    return p


def test_dcn_workspace_and_validation_without_gpu():
    L = _lib.lib()
    p = _dcn_params()
    ws = L.ganet_deform_conv_workspace_bytes(ctypes.byref(p))
    # NCHW input: channel-last copy of x + packed weights
    assert ws >= 2 * 8 * 16 * 64 * 4 + 7 * 64 * 64 * 4
    p2 = _dcn_params(x_layout=1, out_layout=1)
    assert L.ganet_deform_conv_workspace_bytes(ctypes.byref(p2)) >= 7 * 64 * 64 * 4
    assert L.ganet_deform_conv_workspace_bytes(ctypes.byref(p2)) < ws
    # invalid parameters are rejected on the host, before any HIP call
    bad = _dcn_params(kW=0)
    assert L.ganet_deform_conv_workspace_bytes(ctypes.byref(bad)) == 0
    fake = ctypes.c_void_p(256)
    for kw, code in [(dict(kW=0), _lib.GANET_ERR_BAD_SHAPE),
                     (dict(sW=0), _lib.GANET_ERR_BAD_SHAPE),
                     (dict(Cin=63, deformable_groups=2), _lib.GANET_ERR_BAD_SHAPE),
                     (dict(W=2, pW=0), _lib.GANET_ERR_TOO_SMALL),
                     (dict(groups=3), _lib.GANET_ERR_BAD_SHAPE),      # 256 % 3 != 0
                     (dict(dtype=7), _lib.GANET_ERR_BAD_SHAPE)]:
        q = _dcn_params(**kw)
        rc = L.ganet_deform_conv_forward(fake, fake, None, fake, None, fake,
                                         ctypes.byref(q), fake, 1 << 30, None)
        assert rc == code, (kw, rc)
    rc = L.ganet_deform_conv_forward(None, fake, None, fake, None, fake,
                                     ctypes.byref(p), fake, 1 << 30, None)
    assert rc == _lib.GANET_ERR_NULL
    rc = L.ganet_deform_conv_forward(fake, fake, None, fake, None, fake,
                                     ctypes.byref(p), fake, 16, None)
    assert rc == _lib.GANET_ERR_WORKSPACE


def test_algorithm_choice_is_host_logic():
    """ganet_deform_conv_algorithm answers without a GPU: the LFA's 256 -> 64 in bf16 takes the
    contract-then-sample pair; a map whose per-image Y would pass 2 GiB falls back to the fused
    kernel instead of failing; the commuted backward asks for more workspace than the direct."""
    L = _lib.lib()
    lfa = _dcn_params(B=4, Cin=256, H=40, W=100, Cout=64, dtype=_lib.GANET_BF16, x_layout=1,
                      out_layout=1)
    assert L.ganet_deform_conv_algorithm(ctypes.byref(lfa)) == 2
    huge = _dcn_params(B=1, Cin=256, H=1400, W=1400, Cout=64, kW=9, pW=4, dtype=_lib.GANET_BF16,
                       x_layout=1, out_layout=1)      # Y of one image: 2.26 GB
    assert L.ganet_deform_conv_algorithm(ctypes.byref(huge)) == 1
    forced = _dcn_params(B=1, Cin=256, H=1400, W=1400, Cout=64, kW=9, pW=4, dtype=_lib.GANET_BF16,
                         x_layout=1, out_layout=1, algo=2)
    assert L.ganet_deform_conv_algorithm(ctypes.byref(forced)) == 0
    direct = _dcn_params(B=4, Cin=256, H=40, W=100, Cout=64, x_layout=1, out_layout=1, algo=1)
    commuted = _dcn_params(B=4, Cin=256, H=40, W=100, Cout=64, x_layout=1, out_layout=1, algo=2)
    wd = L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(direct))
    wc = L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(commuted))
    assert 0 < wd < wc
    strided = _dcn_params(B=4, Cin=256, H=40, W=100, Cout=64, sW=2, algo=2)
    assert L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(strided)) == 0


def test_small_batch_choice_is_host_logic():
    """Algorithm 5 (taps on different wavefronts) is what one round of workgroups gets - C1's
    batch 1 in bf16 and float32 - and nothing larger; it needs one deformable group and 16-bit
    MFMA operands; the packed weights have one size at every batch size (they carry the
    fragment-major copy whether or not this batch reads it); the kept-Y bit rules it out."""
    L = _lib.lib()
    lfa = dict(Cin=256, H=40, W=100, Cout=64, x_layout=1, out_layout=1)
    for dt in (_lib.GANET_BF16, _lib.GANET_F16, _lib.GANET_F32):
        assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, dtype=dt, **lfa))) == 5
        assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=2, dtype=dt, **lfa))) == 2
        sizes = {L.ganet_deform_conv_packed_weight_bytes(ctypes.byref(_dcn_params(B=b, dtype=dt, **lfa)))
                 for b in (1, 2, 128)}
        assert len(sizes) == 1 and 0 not in sizes
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, mfma_bf16=1, **lfa))) == 5
    # exact float32 cannot run there; an explicit 5 is refused, the automatic choice moves on
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, mfma_bf16=2, **lfa))) in (1, 2)
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, mfma_bf16=2, algo=5, **lfa))) == 0
    assert L.ganet_deform_conv_algorithm(
        ctypes.byref(_dcn_params(B=1, deformable_groups=2, algo=5, **lfa))) == 0
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, deformable_groups=2, **lfa))) == 1
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=128, algo=5, **lfa))) == 5   # on request
    assert L.ganet_deform_conv_algorithm(ctypes.byref(_dcn_params(B=1, algo=6, **lfa))) == 0
    keep = _dcn_params(B=1, algo=_lib.GANET_DCN_ALGO_KEEP_Y, **lfa)
    assert L.ganet_deform_conv_algorithm(ctypes.byref(keep)) == 2
    # two output blocks double the tiles: 250 pixel tiles x 2 is more than one round
    assert L.ganet_deform_conv_algorithm(
        ctypes.byref(_dcn_params(B=1, dtype=_lib.GANET_BF16, **dict(lfa, Cout=128)))) == 2


def test_algo_bits_are_host_logic():
    """GANET_DCN_ALGO_KEEP_Y / GANET_DCN_BWD_OVERWRITE_X ride on `algo`: where the per-tap
    products can be kept the workspace grows to the whole batch's Y and the offset is
    reported; elsewhere the offset is -1 and the forward call refuses the bit."""
    L = _lib.lib()
    none = ctypes.c_size_t(-1).value
    B, H, W, T, Cout = 48, 40, 100, 7, 64
    lfa = dict(B=B, Cin=256, H=H, W=W, Cout=Cout, x_layout=1, out_layout=1)
    plain = _dcn_params(**lfa)
    keep = _dcn_params(algo=_lib.GANET_DCN_ALGO_KEEP_Y, **lfa)
    assert L.ganet_deform_conv_y_offset(ctypes.byref(plain)) == none
    assert L.ganet_deform_conv_y_offset(ctypes.byref(keep)) == 0      # channel-last x: no copy in front
    y_bytes = B * H * W * T * Cout * 4
    wk = L.ganet_deform_conv_workspace_bytes(ctypes.byref(keep))
    wp = L.ganet_deform_conv_workspace_bytes(ctypes.byref(plain))
    assert wk >= y_bytes > 144 << 20 and wp < wk                      # one chunk instead of cache-sized ones
    assert L.ganet_deform_conv_algorithm(ctypes.byref(keep)) == 2
    nchw = _dcn_params(algo=2 | _lib.GANET_DCN_ALGO_KEEP_Y, **dict(lfa, x_layout=0, out_layout=0))
    assert L.ganet_deform_conv_y_offset(ctypes.byref(nchw)) == B * 256 * H * W * 4
    fake = ctypes.c_void_p(256)
    for kw in (dict(dtype=_lib.GANET_BF16), dict(Cout=32), dict(sW=2), dict(mfma_bf16=1),
               dict(algo=1 | _lib.GANET_DCN_ALGO_KEEP_Y), dict(algo=3 | _lib.GANET_DCN_ALGO_KEEP_Y)):
        q = _dcn_params(**dict(dict(lfa, algo=_lib.GANET_DCN_ALGO_KEEP_Y), **kw))
        assert L.ganet_deform_conv_y_offset(ctypes.byref(q)) == none, kw
        rc = L.ganet_deform_conv_forward(fake, fake, None, fake, None, fake, ctypes.byref(q),
                                         fake, 1 << 40, None)
        assert rc == _lib.GANET_ERR_UNSUPPORTED, (kw, rc)
    bad = _dcn_params(algo=0x400, **lfa)
    assert L.ganet_deform_conv_workspace_bytes(ctypes.byref(bad)) == 0
    # the backward ignores the forward's bit, takes its own, refuses unknown ones
    b0 = _dcn_params(**lfa)
    b1 = _dcn_params(algo=_lib.GANET_DCN_ALGO_KEEP_Y | _lib.GANET_DCN_BWD_OVERWRITE_X, **lfa)
    w0 = L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(b0))
    assert w0 > 0 and L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(b1)) == w0
    assert L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(bad)) == 0
    assert L.ganet_deform_conv_backward_y(fake, fake, None, fake, fake, fake, fake, None, fake, None,
                                          ctypes.byref(b0), fake, 1 << 40, None, None) == _lib.GANET_ERR_NULL


def test_extreme_parameters_are_refused_not_overflowed():
    """Sizes whose products would overflow the host arithmetic are refused on the host
    (tools/asan_plan.sh runs the same queries under UBSan with 400 000 random parameter sets)."""
    L = _lib.lib()
    big = 0x7fffffff
    assert L.ganet_dcn_out_size(big, big, 1, 1, 1) == 0          # result beyond int32
    assert L.ganet_dcn_out_size(10, 0, big, big, 1) == 0
    assert L.ganet_dcn_out_size(big, 0, 1, 1, 1) == big
    fake = ctypes.c_void_p(256)
    for kw in (dict(B=big, H=big, W=big), dict(kH=big, kW=big), dict(kW=8192, pW=4096, W=20000),
               dict(pW=big), dict(Cin=1 << 24, Cout=1 << 24), dict(dW=big), dict(sW=big)):
        q = _dcn_params(**kw)
        assert L.ganet_deform_conv_workspace_bytes(ctypes.byref(q)) == 0, kw
        assert L.ganet_deform_conv_backward_workspace_bytes(ctypes.byref(q)) == 0, kw
        rc = L.ganet_deform_conv_forward(fake, fake, None, fake, None, fake, ctypes.byref(q),
                                         fake, 1 << 40, None)
        assert rc in (_lib.GANET_ERR_UNSUPPORTED, _lib.GANET_ERR_BAD_SHAPE,
                      _lib.GANET_ERR_TOO_SMALL), (kw, rc)


def test_decode_validation_without_gpu():
    L = _lib.lib()
    p = _lib.DecodeParams()
    p.B, p.H, p.W = 4, 40, 100
    p.max_kpts, p.max_centers, p.max_groups = 512, 64, 64
    assert L.ganet_decode_workspace_bytes(ctypes.byref(p)) >= 4 * 63 * 8
    out = _lib.DecodeOut()
    fake = ctypes.c_void_p(256)
    rc = L.ganet_decode_batch(fake, fake, fake, ctypes.byref(p), ctypes.byref(out),
                              fake, 1 << 20, None)
    assert rc == _lib.GANET_ERR_NULL          # output pointers are NULL
    for n, _ in _lib.DecodeOut._fields_:
        setattr(out, n, 256)
    p.max_centers = 4096
    rc = L.ganet_decode_batch(fake, fake, fake, ctypes.byref(p), ctypes.byref(out),
                              fake, 1 << 20, None)
    assert rc == _lib.GANET_ERR_UNSUPPORTED
    p.max_centers = 64
    rc = L.ganet_decode_batch(fake, fake, fake, ctypes.byref(p), ctypes.byref(out),
                              fake, 8, None)
    assert rc == _lib.GANET_ERR_WORKSPACE
    # decode + record stream in one call: the same checks, plus the stream and the counters
    need = L.ganet_decode_packed_workspace_bytes(ctypes.byref(p))
    assert need == L.ganet_decode_workspace_bytes(ctypes.byref(p)) + 256
    args = (fake, fake, fake, ctypes.byref(p), ctypes.byref(out))
    assert L.ganet_decode_batch_packed(*args, None, 1 << 20, fake, need, None) == _lib.GANET_ERR_NULL
    assert L.ganet_decode_batch_packed(*args, fake, 8 * 4 - 1, fake, need, None) == _lib.GANET_ERR_WORKSPACE
    assert L.ganet_decode_batch_packed(*args, fake, 1 << 20, fake, need - 1, None) == _lib.GANET_ERR_WORKSPACE


def test_ops_refuse_cpu_tensors():
    import torch
    from ganet_amd import dcn, decode
    x = torch.zeros(1, 64, 8, 16)
    off = torch.zeros(1, 14, 8, 16)
    m = dcn.DeformConv1D(64, 64, 7, 1, 3)
    with pytest.raises(NotImplementedError):
        m(x, off)
    with pytest.raises(NotImplementedError):
        decode.decode_batch(torch.zeros(1, 1, 8, 16), torch.zeros(1, 2, 8, 16),
                            torch.zeros(1, 2, 8, 16))
