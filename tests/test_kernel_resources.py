"""The dominant kernel sits next to a register-allocation cliff ("The compiler as a constraint"): one more value that
lives across its pipelined loop and the allocator spills into scratch -- 120 -> 128 VGPRs, 68 bytes of scratch per lane, 45 -> 97
scalar spills, and 11.5 -> 7.9 M placements/s (measured in round 4 on an innocent-looking extra call in the restart path).  The
results stay exact, so no parity test notices: this one reads the compiler's own resource report."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "usher_amd", "csrc")


def test_the_walk_kernels_do_not_spill(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-structurizecfg-skip-uniform-regions=1",
                        "-I" + os.path.join(ROOT, "include"), "-x", "hip", "-c", os.path.join(CSRC, "ugp_kernels.hip"), "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "k.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = {}
    for b in blocks:
        name = b.split()[0]
        if "k_best8" not in name:
            continue
        vals = {k: int(v) for k, v in re.findall(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", b)}
        seen[name] = vals
    # main walk, LDS-bitmap variant, coarse pass (ARG): <false,false,false,false>, <false,true,false,false>, <false,false,true,false>
    assert len(seen) >= 3, list(seen)
    for name, v in seen.items():
        assert v.get("ScratchSize", 0) == 0, (name, v)
        assert v.get("VGPRs Spill", 0) == 0, (name, v)
        assert v.get("VGPRs", 0) <= 128 and v.get("Occupancy", 4) >= 4, (name, v)     # four waves per SIMD: the grid and the LDS budget assume it
        # (parked kernel arguments and unit-start values: outside the pipelined loop; the third-bound variants -- last template
        # argument true -- park a few more around the question they ask in the restart path)
        assert v.get("SGPRs Spill", 0) <= (80 if name.endswith("Lb1EEEvNS_9Best8ArgsE") else 64), (name, v)
