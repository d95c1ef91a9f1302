#!/usr/bin/env python
"""Lint of the compiled gfx950 ISA for one miscompile of ROCm 7.2's hipcc (EXPERIMENTS.md, "the diagnostic build that lost 5 % of its rays").

At the join block of a divergent `if` the compiler re-enables the lanes with `s_or_b64 exec, exec, sN`.  A VGPR spill store
(or reload) that the register allocator places at the top of such a block must come AFTER that instruction; when a scalar
copy happens to precede the exec restore, the allocator puts the spill code in front of it, where it runs for the lanes of the
`if` body only.  The other lanes later reload stale scratch.  Round 2's -DAMBER_STAMPS build of pt_megakernel had exactly this:

    .LBB10_142:
        s_mov_b64 s[92:93], s[30:31]
        scratch_store_dword off, v71, off offset:64 ; 4-byte Folded Spill      <- Lambertian lanes only
        scratch_store_dword off, v70, off offset:60
        scratch_store_dword off, v64, off offset:56
        s_or_b64 exec, exec, s[8:9]                                            <- Phong lanes come back here
        ...
        scratch_load_dword v64, off, off offset:56 ; 4-byte Folded Reload      <- all lanes: Phong lanes read stale values

This script compiles hip/pt_host.hip to assembly (hipcc cross-compiles without a GPU) with the product's flags plus any
given on the command line, and reports every block in which a `Folded Spill` / `Folded Reload` precedes the block's
`s_or_b64 exec, exec, ...`.      python tools/check_spill_placement.py [extra hipcc flags]      exit status 1 = found."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]


def compile_to_asm(extra):
    out = Path(tempfile.mkdtemp()) / "pt_host.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", *FLAGS, *extra, "--cuda-device-only", "-S", "hip/pt_host.hip", "-o", str(out)],
                   cwd=ROOT / "amber_amd" / "csrc", check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text()


def scan(asm: str):
    """Yields (kernel, block label, line number, text) for spill code that precedes an exec restore inside one block."""
    kernel, block, pending = None, None, []
    for n, line in enumerate(asm.splitlines(), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, block, pending = m.group(1), None, []
            continue
        if re.match(r"^\.LBB\d+_\d+:", line) or "; %bb." in line:
            block, pending = line.split(":")[0].strip(), []
            continue
        text = line.strip()
        if "Folded Spill" in text or "Folded Reload" in text:
            if text.startswith("scratch_") or text.startswith("buffer_"):
                pending.append((n, text))
        elif re.match(r"s_or_b64 exec, exec,", text) or re.match(r"s_or_saveexec_b64", text) and False:
            for pn, pt in pending:
                yield kernel, block, pn, pt
            pending = []
        elif text.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            pending = []


def main():
    extra = sys.argv[1:]
    found = list(scan(compile_to_asm(extra)))
    for kernel, block, n, text in found:
        print(f"{kernel} {block} line {n}: {text}")
    print(f"{len(found)} spill instruction(s) in front of an exec restore" + (f" (flags: {' '.join(extra)})" if extra else ""))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
