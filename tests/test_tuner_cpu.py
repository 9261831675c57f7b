"""The decision logic of the third pruning bound's per-handle tuner (usher_amd/csrc/ugp_tuner.hpp) on the CPU: a small C++ driver plays
the role of ugp_capi.cpp -- it asks for the mode of every sub-batch and reports each finished block's cost per tile -- with costs that
make one mode cheaper; the tuner must try both first, settle on the cheaper one, keep looking at the other now and then (rarely
when the two differ a lot), follow a change, and keep classes of batches apart."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

DRIVER = r"""
#include <cstdio>
#include <cstdlib>
#include "ugp_tuner.hpp"
using ugp::B3Tuner;
// args: n_sub_batches cost_with cost_without [switch_at cost_with2 cost_without2] ; class 0.  Prints the mode of every sub-batch.
int main(int argc, char **argv) {
    const int n = atoi(argv[1]);
    double cw = atof(argv[2]), co = atof(argv[3]);
    const int sw = argc > 4 ? atoi(argv[4]) : -1;
    B3Tuner T;
    bool mode_of_block = true, first = false;
    for (int i = 0; i < n; i++) {
        if (i == sw) { cw = atof(argv[5]); co = atof(argv[6]); }
        uint32_t pos; uint64_t sq;
        const bool m = T.next(0, &pos, &sq);
        if (pos == 0) { mode_of_block = m; first = T.first; }
        if (m != mode_of_block) { printf("mode changed inside a block\n"); return 1; }
        putchar(m ? '1' : '0');
        // (what tuner_poll does when the block's last sub-batch has completed; here without the pipeline's delay)
        if (pos == B3Tuner::kBlock - 1 && !first) T.record(0, m, m ? cw : co);
    }
    putchar('\n');
    // a second class of batches starts its own trial, and does not disturb the first
    uint32_t pos; uint64_t sq;
    printf("%d\n", T.next(2, &pos, &sq) ? 1 : 0);
    printf("%u %u\n", T.blocks[0], T.blocks[2]);
    return 0;
}
"""


def _run(tmp_path, *args):
    src = tmp_path / "tuner_driver.cpp"
    exe = tmp_path / "tuner_driver"
    if not exe.exists():
        src.write_text(DRIVER)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "usher_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)] + [str(a) for a in args], capture_output=True, text=True, check=True)
    return r.stdout.split("\n")


def _blocks(line, k=6):
    assert len(line) % k == 0
    bl = [line[i:i + k] for i in range(0, len(line), k)]
    assert all(b in ("0" * k, "1" * k) for b in bl), bl
    return [b[0] == "1" for b in bl]


def test_trial_then_the_cheaper_mode_with_rare_looks_at_the_other(tmp_path):
    # with the tables 10 % cheaper: on, off, on, off, then on -- except every 16th block
    out = _run(tmp_path, 6 * 70, 1.0, 1.1)
    bl = _blocks(out[0])
    assert bl[:4] == [True, False, True, False]
    later = bl[4:]
    assert sum(later) >= len(later) - len(later) // 16 - 1 and not all(later)
    off = [i + 4 for i, b in enumerate(later) if not b]
    assert all(i % 16 == 15 for i in off), off
    assert out[1] == "1" and out[2].split() == ["70", "1"]
    # without them 10 % cheaper: settles on off, looks at on every 16th block
    bl = _blocks(_run(tmp_path, 6 * 70, 1.1, 1.0)[0])
    later = bl[4:]
    assert sum(later) <= len(later) // 16 + 1 and any(later)
    assert all((i + 4) % 16 == 15 for i, b in enumerate(later) if b)
    # a clear case (2x) is looked at again only every 64th block
    bl = _blocks(_run(tmp_path, 6 * 200, 1.0, 2.0)[0])
    off = [i for i, b in enumerate(bl) if not b and i >= 4]
    assert off and all(i % 64 == 63 for i in off), off


def test_follows_a_change(tmp_path):
    # cheaper with the tables for 40 blocks, then the other way round: the looks at the other mode carry the news
    bl = _blocks(_run(tmp_path, 6 * 140, 1.0, 1.1, 6 * 40, 1.3, 1.0)[0])
    assert sum(bl[4:40]) >= 33
    assert sum(bl[100:]) <= 4, bl[100:]


STATIC_DRIVER = r"""
#include <cstdio>
#include <cstdlib>
#include "ugp_tuner.hpp"
#include "ugp_knobs.hpp"
int main(int argc, char **argv) {
    // the static rule for every (tree shape, batch class), then what the knob parser makes of UGP_BOUND3
    for (int big = 0; big < 2; big++) for (int poly = 0; poly < 2; poly++) { for (int c = 0; c < 3; c++) putchar(ugp::b3_static_choice(poly != 0, c, big ? 10000000ull : 1000000ull) ? '1' : '0'); putchar(' '); }
    printf("%d\n", ugp::Knobs::from_env().bound3);
    // two classes taking turns: no block of either ever completes, nothing is ever recorded
    ugp::B3Tuner T;
    for (int i = 0; i < 600; i++) { uint32_t pos; uint64_t sq; putchar(T.next(i & 1 ? 2 : 0, &pos, &sq) ? '1' : '0'); }
    putchar('\n');
    return 0;
}
"""


def _static(tmp_path, env_value):
    src, exe = tmp_path / "static_driver.cpp", tmp_path / "static_driver"
    if not exe.exists():
        src.write_text(STATIC_DRIVER)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "usher_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if k != "UGP_BOUND3"}
    if env_value is not None:
        env["UGP_BOUND3"] = env_value
    return subprocess.run([str(exe)], capture_output=True, text=True, check=True, env=env).stdout.split("\n")


def test_static_choice_and_the_knob(tmp_path):
    """Round 6: by default the third bound is decided from the tree's shape and the batch's class -- no A/B in
    the caller's steps: every call of a kind runs the same way from the first one on.  UGP_BOUND3=auto keeps the run-time tuner,
    1 / 0 pin it."""
    out = _static(tmp_path, None)
    rule, knob = out[0].split()[:4], out[0].split()[4]
    # (1 M nodes: random shape, polytomy shape; 10 M nodes: the same) x (few rows, tens of rows, hundreds of rows per sample)
    assert rule == ["001", "001", "111", "001"]
    assert knob == "-2"
    assert _static(tmp_path, "auto")[0].split()[4] == "-1"
    assert _static(tmp_path, "1")[0].split()[4] == "1" and _static(tmp_path, "0")[0].split()[4] == "0"


def test_tuner_never_pins_a_mode_it_has_no_figure_for(tmp_path):
    """classes of batches that take turns reopen the tuner's block at every call, so no block completes and nothing is
    recorded; behind the four trial blocks the mode used to stay `with` for good, unmeasured.  It keeps alternating now."""
    modes = _static(tmp_path, "auto")[1]
    tail = modes[100:]
    assert 0.3 < tail.count("1") / len(tail) < 0.7
