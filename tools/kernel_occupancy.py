#!/usr/bin/env python3
"""Registers, spills and the waves per SIMD they allow, for every kernel of the built library objects.

    tools/kernel_occupancy.py            print the table
    tools/kernel_occupancy.py --write    rewrite tests/golden/kernel_occupancy.json from the current build

A convolution tile's rate depends on how many workgroups a CU holds, i.e. on which side of 168 / 128 / 102 ... registers the
compiler lands - and every epilogue kind compiled into a tile moves that number (round 4: two new kinds took the 80-row tile
from 166 to 191 registers, three resident workgroups per CU to two, conv6 / conv10 of a 12-view pass from 541 to 713 us;
no test saw it, the evidence pass did).  tests/test_host_logic.py::test_kernel_occupancy_table compares the build with the
committed table: fewer waves per SIMD or more spilled registers than recorded fail, on the CPU, at build time."""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
TABLE = REPO / "tests" / "golden" / "kernel_occupancy.json"


def waves_per_simd(vgprs: int) -> int:
    """gfx950: 512 vector registers per SIMD lane, allocated in blocks of 8, at most 8 waves."""
    return min(8, 512 // max(8, (vgprs + 7) // 8 * 8))


def object_kernels(obj: Path) -> dict:
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], check=True)
        data = fat.read_bytes()
        out, i, n = {}, 0, 0
        while True:
            j = data.find(b"\x7fELF", i)
            if j < 0:
                break
            k = data.find(b"\x7fELF", j + 4)
            elf = Path(td) / f"co{n}.elf"
            elf.write_bytes(data[j:k if k > 0 else len(data)])
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(elf)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                nm = re.search(r"\.name:\s+(\S+)", blk)
                vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
                if nm and vg and sp:
                    out[nm.group(1)] = {"vgprs": int(vg.group(1)), "spilled": int(sp.group(1)),
                                                       "waves_per_simd": waves_per_simd(int(vg.group(1)))}
            n += 1
            i = j + 4
    return out


def build_table() -> dict:
    table = {}
    for obj in sorted((REPO / "mvlm_amd" / "csrc" / "build").glob("*.o")):
        table.update(object_kernels(obj))
    return dict(sorted(table.items()))


if __name__ == "__main__":
    t = build_table()
    if "--write" in sys.argv:
        TABLE.write_text(json.dumps({k: {"waves_per_simd": v["waves_per_simd"], "spilled": v["spilled"]} for k, v in t.items()}, indent=1) + "\n")
        print(f"{len(t)} kernels -> {TABLE}")
    else:
        for k, v in t.items():
            print(f"{v['vgprs']:4d} regs  {v['waves_per_simd']} waves/SIMD  {v['spilled']:3d} spilled  {k}")
