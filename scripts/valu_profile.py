"""Builds profiles/r0N_valu.json: per frame kernel the counted vector instructions, L1 tag lookups and busy cycles of
the texture-address units per launch (from the summary scripts/gpu_pmc_frame.sh leaves), and the mean issue cost of
the kernel's vector instructions (its compiled instruction mix priced by scripts/isa_issue_cost.py).

    python scripts/valu_profile.py gpurun_out/<prefix>/summary.txt > profiles/r0N_valu.json
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_issue_cost as cost   # noqa: E402

KERNELS = {"setup_kernel": ("raster_kernels.hip", ["-ffp-contract=off"]), "raster_kernel": ("raster_kernels.hip", ["-ffp-contract=off"]),
           "resolve_surface_fast_kernel": ("resolve_kernels.hip", [])}
WANT = {"SQ_INSTS_VALU": "valu_wave_instructions", "TCP_TOTAL_CACHE_ACCESSES_sum": "l1_tag_lookups", "TA_TA_BUSY_sum": "texture_address_busy_cycles",
        "SQ_INSTS_VMEM_RD": "vector_memory_read_instructions"}


def counters(path):
    out, cur = collections.defaultdict(dict), None
    for line in open(path):
        if not line.startswith(" "):
            cur = next((k for k in KERNELS if k + "(" in line or k + "<" in line), None)   # "<": an instantiation of a templated kernel
            continue
        parts = line.split()
        if cur and len(parts) >= 2 and parts[0] in WANT:
            out[cur][WANT[parts[0]]] = float(parts[1])
    return out


def mix(kernel):
    src, extra = KERNELS[kernel]
    with tempfile.TemporaryDirectory() as d:
        lst = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-fno-slp-vectorize", *extra, "-S",
                        "--cuda-device-only", "-o", lst, os.path.join(ROOT, "wgpu-renderer_amd", "csrc", src)], check=True, stderr=subprocess.DEVNULL)
        c, name, first = collections.Counter(), None, None
        for raw in open(lst):
            m = re.match(r"^(_Z\w+):", raw)
            if m:
                name = m.group(1)
                # a templated kernel is compiled once per instantiation ("...kernelILi1024EE..."): the first is counted
                if first is None and re.search(r"\d+" + kernel + "[EI]", name):
                    first = name
                continue
            text = raw.strip()
            if name is not None and name == first and text.startswith("v_"):
                c[cost.classify(text)] += 1
    n = sum(c.values())
    return {"compiled_vector_instructions": dict(c), "issue_cycles_per_instruction": round(sum(cost.CYCLES[k] * v for k, v in c.items()) / n, 3)}


def main():
    table = counters(sys.argv[1])
    doc = {"source": "scripts/gpu_pmc_frame.sh (rocprofv3 counters per dispatch of one 3840x2160 frame at the default view) and scripts/isa_issue_cost.py "
                     "(compiled instruction mix priced with the issue rates of profiles/r02_valu_issue_rates.jsonl)"}
    for k in KERNELS:
        doc[k] = {**table.get(k, {}), **mix(k)}
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
