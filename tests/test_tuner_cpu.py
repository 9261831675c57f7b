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
