"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
declared in include/craft_hip.h, the ctypes table is in sync with the header, and the host-side
helpers mirror the reference's semantics.  No kernels are launched here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "craft_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"^\s*(?:int|const char\*)\s+(craft_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S)


def test_library_builds_and_exports_every_declared_symbol():
    from craft_amd.build import build_extension
    lib_path = build_extension()
    lib = ctypes.CDLL(lib_path)
    fns = declared_functions()
    assert len(fns) >= 20
    for name, _ in fns:
        assert hasattr(lib, name), f"{name} declared in craft_hip.h but not exported by libcraft_hip.so"
    lib.craft_hip_abi_version.restype = ctypes.c_int
    from craft_amd import hip
    header_version = int(re.search(r"#define\s+CRAFT_HIP_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.craft_hip_abi_version() == hip.ABI_VERSION == header_version >= 2
    lib.craft_hip_error_string.restype = ctypes.c_char_p
    assert b"alignment" in lib.craft_hip_error_string(10002)


def test_ctypes_table_matches_header():
    from craft_amd import hip
    decl = dict(declared_functions())
    for name, sig in hip._SIGS.items():
        assert name in decl, f"{name} bound in hip.py but not declared in the header"
        params = [p.strip() for p in decl[name].replace("\n", " ").split(",")]
        assert len(params) == len(sig), f"{name}: header has {len(params)} parameters, ctypes table {len(sig)}"
        for p, t in zip(params, sig):
            if "*" in p:
                assert t is ctypes.c_void_p, f"{name}: '{p}' should be a pointer"
            elif p.startswith("long"):
                assert t is ctypes.c_long, f"{name}: '{p}' should be long"
            elif p.startswith("float"):
                assert t is ctypes.c_float, f"{name}: '{p}' should be float"
            elif p.startswith("int"):
                assert t is ctypes.c_int, f"{name}: '{p}' should be int"
    missing = set(decl) - set(hip._SIGS) - {"craft_hip_abi_version", "craft_hip_error_string"} - hip.HOST_FUNCTIONS
    assert not missing, f"declared but unbound: {missing}"


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from craft_amd import CRAFT, default_args, hip
    m = CRAFT(default_args()).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), iters=1)
    with pytest.raises(hip.CraftHipError):
        hip.call("craft_tokens_to_nchw", torch.zeros(1, 4, 4), 4, 1, 4, 4, torch.zeros(1, 4, 2, 2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "craft_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "from oracle" not in src and "import oracle" not in src, f"{f} imports the oracle"


def test_input_padder_semantics():
    """utils.py:14-31: 'sintel' pads symmetrically, other modes pad the bottom; unpad inverts pad."""
    from craft_amd import InputPadder
    x = torch.arange(2 * 3 * 436 * 1022, dtype=torch.float32).reshape(2, 3, 436, 1022)
    p = InputPadder(x.shape)
    (y,) = p.pad(x)
    assert y.shape[-2:] == (440, 1024) and p._pad == [1, 1, 2, 2]
    assert torch.equal(p.unpad(y), x)
    k = InputPadder((375, 1242), mode="kitti")
    assert k._pad == [3, 3, 0, 1]
    assert InputPadder((448, 1024))._pad == [0, 0, 0, 0]


def test_args_roundtrip_mutation():
    """The ctor writes corr_levels / corr_multiplier / *_trans_config back into args (network.py:33,57,92,106,127)."""
    from craft_amd import CRAFT, default_args
    a = default_args()
    CRAFT(a)
    assert a.corr_levels == 4 and a.corr_multiplier == 1
    assert a.inter_trans_config.out_attn_scores_only and a.intra_trans_config.out_attn_probs_only
    assert a.f2_trans_config.has_input_skip and a.inter_trans_config.tie_qk_scheme == "shared"
    b = default_args(corr_radius=-1)
    CRAFT(b)
    assert b.corr_radius == 4


def test_tied_inter_frame_projection():
    from craft_amd import CRAFT, default_args
    m = CRAFT(default_args())
    assert m.corr_fn.setrans.key.weight is m.corr_fn.setrans.query.weight
    assert m.corr_fn.setrans.key.bias is m.corr_fn.setrans.query.bias
    assert sum(p.numel() for p in m.parameters()) == 6307435      # logs/11 craft-chairs-f2full-110621.txt:36
