import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "refshim: needs /root/reference (build container only)")


def built(cases):
    """parametrize lists filtered to what the loaded library compiles: the measured-but-not-selected tile variants and kernel modes exist only
    in a -DIMH_EXPERIMENTAL build (IMH_EXPERIMENTAL=1 python -m imagharmony_amd.build; imagharmony_amd.lib.EXP_VARIANTS)"""
    from imagharmony_amd import lib as L

    def cfg_of(c):
        if isinstance(c, dict):
            return c.get("cfg")
        if isinstance(c, tuple) and c and isinstance(c[0], tuple):
            return c[0]
        return c if isinstance(c, tuple) else None
    return [c for c in cases if cfg_of(c) is None or len(cfg_of(c)) < 2 or L.variant_built(cfg_of(c))]


def experimental():
    from imagharmony_amd import lib as L
    return L.experimental()


def rel_rms(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float(((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)))


# Measured parity numbers on record: every full-size / golden comparison appends {name: {value, bound, ...}} here, the
# round's copy is committed as profiles/rNN_parity.json (gpurun_out/ is merged back from the GPU box).
PARITY_JSON = os.environ.get("IMH_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity_measured.json")


def record_parity(name, value, bound, **extra):
    import json
    try:
        os.makedirs(os.path.dirname(PARITY_JSON), exist_ok=True)
        try:
            with open(PARITY_JSON) as f:
                d = json.load(f)
        except (OSError, ValueError):
            d = {}
        d[name] = {"rel_rms": float(value), "bound": float(bound), **extra}
        with open(PARITY_JSON, "w") as f:
            json.dump(d, f, indent=1, sort_keys=True)
    except OSError:
        pass


def cmp_cfg_golden(y, g, case, what, tol, name=None):
    """y [B, L, C] (any float dtype / device) against a cfg-shape fixture entry (oracle/gen_golden.py compress():
    sampled rows in fp16 + row / column sums of the whole reference output)"""
    from oracle.gen_golden import sample_rows
    y = y.float().cpu()
    e = g[what]
    idx = sample_rows(case)
    r_rows = rel_rms(y[:, idx], e["rows"].float())
    r_rs = rel_rms(y.double().sum(-1), e["row_sum"])
    r_cs = rel_rms(y.double().sum(1), e["col_sum"])
    if name:
        record_parity(name, r_rows, tol, row_sum_rel_rms=r_rs, col_sum_rel_rms=r_cs)
    assert r_rows < tol, f"{case} {what}: sampled rows rel-rms {r_rows:.3e}"
    assert r_rs < 2 * tol and r_cs < 2 * tol, f"{case} {what}: row / column sums rel-rms {r_rs:.3e} / {r_cs:.3e}"
    return r_rows


def ref_row_stats(x, slots):
    """row statistics of x [rows, C] in the hand-over format of csrc/imh_lnstats.h, computed with torch in fp64:
    [rows, slots, 2] fp32 = (sum, sum of squared deviations from the slot mean) of every run of C / slots channels"""
    import torch
    rows, C = x.shape
    v = x.double().view(rows, slots, C // slots)
    s = v.sum(-1)
    m2 = (v - v.mean(-1, keepdim=True)).pow(2).sum(-1)
    return torch.stack([s, m2], -1).float().contiguous()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
