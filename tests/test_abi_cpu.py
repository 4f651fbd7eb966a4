"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header
declares, and the Python mirror keeps the reference's state-dict contract."""
import ctypes
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "srhip.h")).read()
    return sorted(set(re.findall(r"SRHIP_API\s+[\w\s\*]+?\b(srhip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from superresolution_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"libsrhip.so does not export {n}"
    # and the ctypes table covers exactly the header
    assert sorted(_lib.EXPORTED) == names
    assert _lib.lib.srhip_version() >= 1


def test_conv_args_struct_layout():
    from superresolution_amd._lib import Conv3x3Args
    # 6 pointers + 14 4-byte scalars, then the sign-bit fields: pointer, int (+4 padding), pointer
    assert ctypes.sizeof(Conv3x3Args) == 6 * 8 + 14 * 4 + 8 + 8 + 8
    assert Conv3x3Args.aux.offset == 104 and Conv3x3Args.aux_mode.offset == 112 and Conv3x3Args.dst2.offset == 120
    assert Conv3x3Args.n.offset == 48 and Conv3x3Args.dst_nchw_f32.offset == 48 + 10 * 4 and Conv3x3Args.res_mode.offset == 92 and Conv3x3Args.ca_stride.offset == 96 and Conv3x3Args.cout_w.offset == 100


def test_packed_bytes():
    from superresolution_amd._lib import lib
    assert lib.srhip_conv3x3_packed_bytes(64, 64) == 9 * 4 * 2 * 64 * 16
    assert lib.srhip_conv3x3_packed_bytes(96, 64) == 9 * 6 * 2 * 64 * 16
    assert lib.srhip_conv3x3_packed_bytes(16, 3) == 9 * 1 * 1 * 64 * 16
    assert lib.srhip_conv3x3_signbits_bytes(2, 3, 77) == 2 * 3 * 3 * 64 * 4      # 1 bit per element, rows padded to 32 pixels
    assert lib.srhip_conv3x3_signbits_bytes(1, 1024, 1024) * 8 == 1024 * 1024 * 64


def test_supersampler_state_dict_contract(ss_golden):
    from superresolution_amd import SuperSampler
    z, sd = ss_golden
    m = SuperSampler(3, 10, 1)
    own = m.state_dict()
    assert list(own.keys()) == [str(k) for k in z["keys"]]
    for k, v in own.items():
        assert tuple(v.shape) == sd[k].shape, k
        assert v.dtype == torch.float32
    assert sum(p.numel() for p in m.parameters()) == 913155
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)


def test_supersampler_init_matches_reference_seed_for_seed(ss_golden):
    from superresolution_amd import SuperSampler
    z, _ = ss_golden
    torch.manual_seed(42)
    m = SuperSampler(3, 10, 1)
    for name, p in m.named_parameters():
        ref = z["init." + name]
        got = p.detach().numpy().ravel()
        assert np.array_equal(got[:8], ref[:-1]), name
        assert np.float32(p.detach().double().sum().item()) == ref[-1], name


def test_ops_refuse_cpu_tensors():
    import pytest
    from superresolution_amd import functional as F_
    with pytest.raises(ValueError):
        F_.nchw_f32_to_nhwc_bf16(torch.zeros(1, 3, 4, 4))
    with pytest.raises(ValueError):
        F_.pixel_shuffle(torch.zeros(1, 4, 2, 2), 2)


def _header_struct_fields(name: str):
    """(type, field) pairs of `typedef struct { ... } name;` in include/srhip.h, comments stripped, arrays as 'x[3]'."""
    text = open(os.path.join(ROOT, "include", "srhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct \{([^}]*)\}\s*" + name + r"\s*;", text).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const void\*|void\*|const float\*|float\*|long long|int|float)\s+(.*)", decl)
        assert m, decl
        for fld in m.group(2).split(","):
            out.append((m.group(1), fld.strip()))
    return out


def _expected_layout(fields):
    """Offsets and size the C compiler gives these fields on x86-64 / LP64 (natural alignment)."""
    off, offs, align_max = 0, {}, 1
    for typ, fld in fields:
        size = 8 if ("*" in typ or typ == "long long") else 4
        cnt = 1
        m = re.match(r"(\w+)\[(\d+)\]", fld)
        if m:
            fld, cnt = m.group(1), int(m.group(2))
        off = (off + size - 1) // size * size
        offs[fld] = off
        off += size * cnt
        align_max = max(align_max, size)
    return offs, (off + align_max - 1) // align_max * align_max


def test_every_args_struct_matches_the_header_field_for_field():
    from superresolution_amd import _lib
    for cname, cls in (("srhip_conv3x3_args", _lib.Conv3x3Args), ("srhip_conv3x3_pair_args", _lib.Conv3x3PairArgs),
                       ("srhip_conv_gemm_args", _lib.ConvGemmArgs), ("srhip_conv1x1_chain_args", _lib.Conv1x1ChainArgs),
                       ("srhip_wgrad_job", _lib.WgradJob)):
        fields = _header_struct_fields(cname)
        offs, size = _expected_layout(fields)
        names = [re.sub(r"\[\d+\]", "", f) for _, f in fields]
        assert [f[0] for f in cls._fields_] == names, cname
        assert ctypes.sizeof(cls) == size, cname
        for fld, off in offs.items():
            assert getattr(cls, fld).offset == off, (cname, fld)
    assert ctypes.sizeof(_lib.Conv3x3Args) == 128 and ctypes.sizeof(_lib.Conv3x3PairArgs) == 104
    assert ctypes.sizeof(_lib.ConvGemmArgs) == 136 and ctypes.sizeof(_lib.Conv1x1ChainArgs) == 120
    assert ctypes.sizeof(_lib.WgradJob) == 64


def integration_snippet() -> str:
    """The ctypes example of INTEGRATION.md, verbatim."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(import ctypes, os, torch.*?)```", text, flags=re.S)
    assert m, "INTEGRATION.md lost its ctypes example"
    return m.group(1)


def test_integration_stub_declares_the_same_struct():
    from superresolution_amd._lib import Conv3x3Args
    src = integration_snippet()
    cls_src = src[src.index("class Conv3x3Args"):src.index("lib.srhip_conv3x3_packed_bytes.argtypes")]
    ns = {"ctypes": ctypes}
    exec(cls_src, ns)           # the class statement and its sizeof assertion only: no library call
    doc = ns["Conv3x3Args"]
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in Conv3x3Args._fields_]
    assert ctypes.sizeof(doc) == ctypes.sizeof(Conv3x3Args)
    # and the call in the example initialises every field
    call = src[src.index("a = Conv3x3Args("):src.index("if lib.srhip_conv3x3_fwd")]
    assert call.count(",") + 1 == len(Conv3x3Args._fields_)
