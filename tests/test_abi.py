"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/pyipm_newton.h declares; the product path fails loudly without a GPU (no CPU fallback);
nothing under pyipm_amd/ imports the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="pyipm_newton.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pyipm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from pyipm_amd import newton
    names = _header_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(newton.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(newton.exported_symbols()) == names, "ctypes binding and header disagree"


def test_library_exports_every_lbfgs_symbol():
    """include/pyipm_lbfgs.h: same shared object, its own ctypes table."""
    from pyipm_amd import newton, lbfgs
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["pyipm_lbfgs.h", "pyipm_newton.h"]
    names = _header_functions("pyipm_lbfgs.h")
    assert len(names) == 10
    lib = ctypes.CDLL(newton.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(lbfgs.exported_symbols()) == names, "ctypes binding and header disagree"
    b = lbfgs.load().pyipm_lbfgs_workspace_bytes(262144, 1024, 3072, 9, 256)
    assert b > 262144 * 4096 * 8                               # the padded Jacobian operand dominates
    assert lbfgs.load().pyipm_lbfgs_workspace_bytes(100, 0, 0, 33, 256) == 0       # max_pairs <= 32


def test_workspace_query_needs_no_gpu():
    from pyipm_amd import newton
    lib = newton.load_library()
    b1 = lib.pyipm_newton_workspace_bytes(16384, 4096, 6144, 256, 1, 0)
    assert b1 > 32768 * 32768 * 8
    b8 = lib.pyipm_newton_workspace_bytes(16384, 4096, 6144, 256, 8, 3)
    assert b8 < b1 / 4
    assert lib.pyipm_newton_workspace_bytes(10, 0, 0, 96, 1, 0) == 0      # nb must be a multiple of 128
    assert lib.pyipm_newton_workspace_bytes(10, 0, 0, 256, 2, 2) == 0     # rank out of range


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyipm_amd.newton import NewtonCore, NewtonError
    from pyipm_amd.lbfgs import LbfgsCore
    with pytest.raises(NewtonError):
        NewtonCore(3, 1, 3)
    with pytest.raises(NewtonError):
        LbfgsCore(3, 1, 3, 5)


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyipm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt.replace(
                        "``/root/reference", "").replace("/root/reference/pyipm.py:", ""):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_bench_watchdog_prints_an_error_line_instead_of_hanging(tmp_path):
    """bench.py's watchdog: a phase that overruns -- a stalled collective, a communicator that never comes
    up -- ends the run with ONE JSON line carrying an `error` field on the descriptor bench.py prints its result to, and
    exit status 3; a phase that finishes in time leaves nothing behind."""
    import json
    import subprocess
    import sys
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "wd = bench.Watchdog(1, 0, {'metric': 'newton_steps_per_sec', 'n_gpus': 8})\n"
        "wd.watch('quick phase', 30); time.sleep(0.2); wd.clear()\n"
        "wd.watch('communicator bring-up', 1.5)\n"
        "time.sleep(30)\n"
        "print('not reached')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in p.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["stalled_phase"] == "communicator bring-up" and "watchdog" in d["error"]
    assert d["metric"] == "newton_steps_per_sec" and d["n_gpus"] == 8
    # other ranks exit silently
    p2 = subprocess.run([sys.executable, "-c", code.replace("Watchdog(1, 0,", "Watchdog(1, 3,")], capture_output=True, text=True, timeout=60)
    assert p2.returncode == 3 and not [l for l in p2.stdout.splitlines() if l.startswith("{")]


def test_bench_ladder_rung_that_raises_keeps_the_best_completed_number():
    """bench.py's ladder: a rung whose step RAISES -- an error code from the library -- ends the run like a rung
    that stalls: ONE line with the best completed rung's value and status 0; without a completed rung an `error` line and status 3."""
    import json
    import subprocess
    import sys
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "wd = bench.Watchdog(1, 0, {'metric': 'newton_steps_per_sec', 'n_gpus': 8})\n"
        "class Core:\n"
        "    def set_option(self, k, v): pass\n"
        "    def comm_ranks(self): return 8\n"
        "    def comm_bcast_mode(self): return 0\n"
        "n = [0]\n"
        "def one_step():\n"
        "    n[0] += 1\n"
        "    if n[0] > FAIL_AFTER: raise RuntimeError('pyipm_newton_step_dist failed (-6): tile chain: a poll timed out')\n"
        "    time.sleep(0.01)\n"
        "bench.run_ladder(Core(), one_step, lambda: None, wd, 8, 0, 2, 1, lambda x: x, {'config': {'kkt_dim': 1}})\n"
        "print('not reached')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code.replace("FAIL_AFTER", "7")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr[-500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in p.stdout
    d = json.loads(lines[0])
    assert d["value"] > 0 and len(d["ladder"]) == 2 and "rung 2" in d["ladder_failed_at"] and "poll timed out" in d["ladder_failure"]
    assert d["wire_form"] in [r["wire_form"] for r in d["ladder"]] and "error" not in d
    p = subprocess.run([sys.executable, "-c", code.replace("FAIL_AFTER", "0")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and "rung 0" in d["failed_phase"] and "error" in d


def test_bench_default_panel_width():
    """bench.py's panel width when --nb is not given: 256 on one GPU; across GPUs 256 while the owners' chain is the step
    and 1024 where the bulk update is (rank replays at three widths, profiles/r04_z_replay_nb.txt)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_panel_width(1, 32768) == 256 and bench.default_panel_width(1, 131072) == 256
    assert bench.default_panel_width(8, 32768) == 256 and bench.default_panel_width(2, 40960) == 256
    assert bench.default_panel_width(8, 131072) == 1024 and bench.default_panel_width(2, 65536) == 1024


def src_all():
    out = ""
    d = os.path.join(ROOT, "pyipm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            out += open(os.path.join(d, f)).read()
    return out


def test_option_lists_of_header_and_sources_agree():
    """include/pyipm_newton.h documents 28 public options and names the 12 expert ones -- at most 40 names in all, 15 expert
    (round 6 removed 35 switches and the code behind the ones whose measurements lost): both lists against
    the option names the sources actually compare with, and against the table that gates the expert ones -- without a GPU."""
    hdr = open(os.path.join(ROOT, "include", "pyipm_newton.h")).read()
    pub = re.search(r"PUBLIC OPTIONS:(.*?)\n \*\n", hdr, re.S).group(1)
    exp = re.search(r"EXPERT OPTIONS:(.*?)\n \*   \(which stream", hdr, re.S).group(1)
    names = lambda blk: [w for w in re.sub(r"[*\n]", " ", blk).replace(",", " ").split() if w]     # noqa: E731
    pub, exp = names(pub), names(exp)
    assert len(pub) + len(exp) <= 40 and len(exp) <= 15 and len(set(pub)) == len(pub) and len(set(exp)) == len(exp) and not set(pub) & set(exp)
    for gone in ("early_head", "early_first", "fused_head", "fused_head_rows", "head_split", "head_split_rows", "head_serial", "tile_ny3",
                 "bulk_bn_all", "asm_split", "inpanel32", "s_across"):
        assert gone not in pub and gone not in exp and ("ctx->" + gone) not in src_all(), gone
    src = ""
    for f in ("pyipm_newton.hip", "dist_impl.hpp"):
        src += open(os.path.join(ROOT, "pyipm_amd", "csrc", f)).read()
    accepted = set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', src))
    assert accepted == set(pub) | set(exp), accepted ^ (set(pub) | set(exp))
    table = re.search(r"kExpert\[\] = \{(.*?)\};", src, re.S).group(1)
    assert set(re.findall(r'"([a-z_0-9]+)"', table)) == set(exp)
