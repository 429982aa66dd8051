"""Kernel-variant selection for the GEMM entry points -- deterministic by default.

Every variant of vlp_gemm_nt computes the same contraction with every output element's fp32 chain walked in ascending k: the NT variants
are BIT-IDENTICAL to one another (measured in round 4 on the step's shapes, tests/test_00_kernels_gpu.py::test_gemm_nt_variant_identity,
profiles/r04_nt_variant_identity.json); they differ in speed only.  vlp_gemm_tn's split-M factor does change the fp32 summation order of a
weight gradient (low-order bits).  Round 1 chose variants by timing candidates on first use, which made speed, HBM traffic and -- for the
wgrads -- the low-order bits depend on the box and on timer noise.  Now the choice is a pure function of the problem shape:

  1. `VLP_NT_VARIANT` / `VLP_TN_CHOICE` environment overrides (A/B runs);
  2. the committed table `vlp_amd/tuned_gfx950.json` (measured once on an MI355X with `python -m vlp_amd.tuning --tune`,
     i.e. `VLP_AUTOTUNE=1`; the file travels with the source, so every box runs the same kernels);
  3. a shape heuristic for everything the table does not list.

`VLP_AUTOTUNE=1` re-enables the timing search (engine.py); `dump()` writes what it found so it can be committed.
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE_PATH = os.environ.get("VLP_TUNE_TABLE") or os.path.join(_HERE, "tuned_gfx950.json")
AUTOTUNE = os.environ.get("VLP_AUTOTUNE", "0") == "1"

SKINNY_SPLITS = (2, 3, 4, 6, 8, 12, 16)
TN_SPLIT_CANDIDATES = (0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16)

_table = None
_found = {}          # what an autotune run measured in this process: key string -> choice


def _load():
    global _table
    if _table is None:
        _table = {}
        if os.path.exists(TABLE_PATH):
            with open(TABLE_PATH) as f:
                _table = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    return _table


def _key(kind, M, N, K):
    return "%s:%d,%d,%d" % (kind, M, N, K)


def lookup(kind, M, N, K):
    v = _load().get(_key(kind, M, N, K))
    if isinstance(v, list):
        v = tuple(v)
    return v


def remember(kind, M, N, K, choice):
    _found[_key(kind, M, N, K)] = list(choice) if isinstance(choice, tuple) else choice


def dump(path):
    """Merge this process's autotune results into `path` (json)."""
    cur = {}
    if os.path.exists(path):
        with open(path) as f:
            cur = json.load(f)
    cur.update(_found)
    with open(path, "w") as f:
        json.dump(cur, f, indent=0, sort_keys=True)
    return cur


# ---- heuristics ------------------------------------------------------------------------------------------------------
def nt_heuristic(M, N, K):
    """vlp_gemm_nt variant (include/vlp_hip.h): +8 = XCD-aware tile order, +16 = LDS-DMA ring (raw barrier, counted vmcnt), 64 + cfg = wave-pipelined family.
    Measured with cold operands (tools/nt_lab.py --rotate=12), which is what a training step sees: the rings win every shape."""
    if M <= 1024:
        return 1                      # few workgroups: 128x128 LDS-DMA double buffer
    if N <= 1024:
        return 77                     # 256x128 tiles, wave-pipelined 3-slot ring (gemm_nt_wp.hip): narrow outputs (252 workgroups at M = 10 688, N = 768)
    # Packed (padding-free) steps run every GEMM at M' = sum of the kept lengths, a different value each step (7 000 .. 10 688 at
    # B = 64): no table row can name it, so the rules below carry what tools/varlen_lab.py measured over that range
    # (profiles/r05_varlen_m_sweep.txt, cold operands).
    if M >= 6144 and 1024 < N <= 2560 and N % 128 == 0 and K > 512:
        return 264                    # QKV forward: persistent k-stream kernel (52.5 us at M = 10 688 down to 44.4 at 7 936; rings 57.7 -> 46.3)
    return 29                         # 256x256 tiles, 2-stage ring


def tn_heuristic(M, N, K):
    """(variant, split-M factor) of vlp_gemm_tn for C[N,K] = A[M,N]^T B[M,K]: tiles x splits should fill 256 CUs x 2."""
    if M < 1024:
        return (2, 0)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    want = max(1, 512 // tiles)
    best = 0
    for s in TN_SPLIT_CANDIDATES:
        if 1 < s <= want and M // s >= 128:
            best = s
    return (2, best)


def skinny_heuristic(M, N, K):
    """('v', variant) | ('s', splits) for the decoder's M <= 1024 GEMMs (engine._nt_skinny)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    best = None
    for s in SKINNY_SPLITS:
        if s * 2 <= K // 64 and tiles * s <= 512:
            best = s
    if tiles >= 128 or best is None:
        return ("v", 1)
    return ("s", best)


def _nt_overrides():
    """VLP_NT_OVERRIDE="M,N,K=variant;M,N,K=variant" -- per-shape A/B runs inside the real step (same box, same process layout)."""
    out = {}
    for item in os.environ.get("VLP_NT_OVERRIDE", "").split(";"):
        if "=" in item:
            k, v = item.split("=")
            out[tuple(int(x) for x in k.split(","))] = int(v)
    return out


def _nt_rules():
    """VLP_NT_RULES="N,K,Mlo,Mhi=variant;..." -- A/B runs of a variant over a RANGE of row counts (packed steps change M every step)."""
    out = []
    for item in os.environ.get("VLP_NT_RULES", "").split(";"):
        if "=" in item:
            k, v = item.split("=")
            n, kk, lo, hi = (int(x) for x in k.split(","))
            out.append((n, kk, lo, hi, int(v)))
    return out


def nt_variant(M, N, K):
    env = os.environ.get("VLP_NT_VARIANT")
    if env:
        return int(env)
    for n, kk, lo, hi, v in _nt_rules():
        if n == N and kk == K and lo <= M <= hi:
            return v
    ov = _nt_overrides().get((M, N, K))
    if ov is not None:
        return ov
    v = lookup("nt", M, N, K)
    return int(v) if v is not None else nt_heuristic(M, N, K)


def tn_choice(M, N, K):
    env = os.environ.get("VLP_TN_CHOICE")
    if env:
        a, b = env.split(",")
        return (int(a), int(b))
    v = lookup("tn", M, N, K)
    return (int(v[0]), int(v[1])) if v is not None else tn_heuristic(M, N, K)


def skinny_choice(M, N, K):
    v = lookup("sk", M, N, K)
    return (str(v[0]), int(v[1])) if v is not None else skinny_heuristic(M, N, K)


def _dump_at_exit():
    path = os.environ.get("VLP_TUNE_DUMP")
    if path and _found:
        dump(path)


import atexit      # noqa: E402
atexit.register(_dump_at_exit)
