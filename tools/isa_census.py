#!/usr/bin/env python3
"""Instruction census of one kernel of hevc_amd/csrc/kernels.hip from the compiler's gfx950 assembly.

    python3 tools/isa_census.py [--kernel k_inter_tileIhLi64ELi512ELb0] > profiles/r3_isa_census_tile64.txt

Compiles the kernels with line tables (hipcc -gline-tables-only, device code only; no GPU needed), disassembles the
kernel and counts its instructions by class and by source region.  An instruction belongs to the region of the line of
the kernel's own body it was compiled from: for a helper inlined from elsewhere in the file that is the line of the
call (llvm-symbolizer --inlines gives the chain).  The counts are STATIC: every path of the kernel once, loops once.  What a
wave executes depends on its role (which items it owns) and on the region's motion vectors; the per-wave figures of the
hardware counters (SQ_INSTS_VALU / SQ_WAVES and so on, tools/pmc_passes.sh) are the dynamic side of the same census.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "hevc_amd", "csrc", "kernels.hip")

# (class, regular expression on the mnemonic), first match wins
CLASSES = [
    ("dot product", r"v_dot"),
    ("matrix core", r"v_mfma|v_smfma"),
    ("byte align / permute", r"v_alignb|v_perm_b32"),
    ("bias (xor)", r"v_xor"),
    ("pack / field (shift-or, and, bfe, sdwa moves)", r"v_lshl_or|v_and_or|v_or3|v_and_b32|v_or_b32|v_bfe|v_bfi|v_lshlrev_b32_sdwa|v_cndmask_b32_sdwa|v_mov_b32_sdwa|v_lshrrev_b16|v_lshlrev_b16"),
    ("shift", r"v_ashrrev|v_lshrrev|v_lshlrev"),
    ("add / sub / mul", r"v_add_|v_sub|v_mul|v_mad_[iu]32|v_mad_[iu]16|v_add3|v_lshl_add_u32|v_add_lshl"),
    ("clip (min / max / med3)", r"v_min|v_max|v_med3"),
    ("compare / select", r"v_cmp|v_cndmask"),
    ("address (64-bit)", r"v_lshl_add_u64|v_mad_[iu]64|v_add_co|v_addc"),
    ("move / broadcast", r"v_mov|v_readfirstlane|v_readlane|v_writelane|v_accvgpr|v_swap|v_mbcnt|v_bcnt"),
    ("other vector ALU", r"v_"),
    ("LDS", r"ds_"),
    ("global memory", r"global_|buffer_|flat_|scratch_"),
    ("scalar memory", r"s_load|s_buffer_load"),
    ("wait / barrier / nop", r"s_waitcnt|s_barrier|s_nop|s_sleep"),
    ("scalar branch", r"s_cbranch|s_branch|s_endpgm|s_setpc|s_getpc|s_swappc"),
    ("scalar ALU", r"s_"),
]

# source regions of k_inter_tile (line ranges are looked up by the markers below, not hard-coded)
TILE_MARKS = [
    ("entry, region description", r"template <typename pel_t, int TS, int NTHR, bool WP>"),
    ("H: window fetch", r"unsigned w\[2\]\[HIT\]\[2\]\[WD\];"),
    ("H: filter set-up, item decode", r"const unsigned \*lt = kLumaTab\[t\.mvx\[l\] & 3\]"),
    ("H: 16-bit samples", r"if constexpr \(W16\) \{"),
    ("H: whole-sample position", r"\} else if \(\(chroma \? t\.mvx\[l\] & 7 : t\.mvx\[l\] & 3\) == 0\) \{"),
    ("H: luma 8-tap", r"\} else if \(!chroma\) \{"),
    ("H: chroma 4-tap", r"for \(int i = 0; i < 3; i\+\+\) x\[i\] = w\[l\]\[it\]\[rr\]\[i\]"),
    ("H: pack and store to LDS", r"uint4 \*dst = chroma \?"),
    ("barrier, V set-up (rounding, weights)", r"#undef MT_DECODE"),
    ("V: luma", r"const int q = v % \(TS / 4\), rp = v / \(TS / 4\);"),
    ("V: luma round, clip, store", r"if \(v < LT && rx \+ 4 \* q < s\.w\) \{"),
    ("V: chroma", r"const int vc = v - LT;"),
]


def regions_of_tile(src_lines):
    """[(first_line, name)], ascending; the last region ends with the kernel."""
    out = []
    start = 0
    for name, pat in TILE_MARKS:
        for i in range(start, len(src_lines)):
            if re.search(pat, src_lines[i]):
                out.append((i + 1, name))
                start = i + 1
                break
        else:
            raise SystemExit("marker not found in kernels.hip: " + pat)
    return out


def classify(mnemonic):
    for name, pat in CLASSES:
        if re.match(pat, mnemonic):
            return name
    return "other"


LLVM = "/opt/rocm/lib/llvm/bin/"


def device_elf(hipcc, workdir):
    """kernels.hip compiled for gfx950 with line tables (which name the call sites of inlined helpers too)."""
    co, elf = os.path.join(workdir, "kernels.co"), os.path.join(workdir, "kernels.elf")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "--offload-device-only", "-c", SRC, "-o", co],
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], check=True)
    return elf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_inter_tileIhLi64ELi512ELb0", help="substring of the mangled kernel name")
    ap.add_argument("--elf", help="device code object made earlier (see device_elf())")
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    a = ap.parse_args()
    tmp = tempfile.TemporaryDirectory()
    elf = a.elf or device_elf(a.hipcc, tmp.name)
    syms = subprocess.run([LLVM + "llvm-readelf", "-sW", elf], check=True, capture_output=True, text=True).stdout
    name = next((l.split()[-1] for l in syms.splitlines() if " FUNC " in l and a.kernel in l), None)
    if name is None:
        raise SystemExit("no kernel matching " + a.kernel)
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + name, elf], check=True, capture_output=True, text=True).stdout
    insts = []  # (address, mnemonic)
    for l in dis.splitlines():
        m = re.match(r"\s+(\w+)\b.*//\s*([0-9A-Fa-f]{8,}):", l)
        if m:
            insts.append((int(m.group(2), 16), m.group(1)))
    sym = subprocess.run([LLVM + "llvm-symbolizer", "--obj=" + elf, "--inlines"], input="".join("0x%x\n" % ad for ad, _ in insts),
                         check=True, capture_output=True, text=True).stdout
    chains = [c.strip().splitlines() for c in sym.strip().split("\n\n")]
    if len(chains) != len(insts):
        raise SystemExit("symbolizer answered %d of %d addresses" % (len(chains), len(insts)))
    src_lines = open(SRC).read().splitlines()
    tile = "k_inter_tile" in name
    regs = regions_of_tile(src_lines) if tile else [(1, "kernel")]

    def region(line):
        cur = regs[0][1]
        for lo, nm in regs:
            if line >= lo:
                cur = nm
        return cur

    # the clamped fetch of a window row (regions at a picture border only) is told apart by the helper's own lines
    slow_lo = next(i + 1 for i, l in enumerate(src_lines) if "pins the clamping arithmetic of this rare path" in l)
    slow_hi = next(i + 1 for i in range(slow_lo, len(src_lines)) if src_lines[i].startswith("}"))
    counts = collections.defaultdict(lambda: collections.Counter())
    mnems = collections.defaultdict(lambda: collections.Counter())
    order = []
    for _, nm in regs:
        order.append(nm)
        if nm == "H: window fetch":
            order.append("H: window fetch, clamped (picture border)")
    for (ad, mn), ch in zip(insts, chains):
        # frames come innermost first: function, file:line:col, function, file:line:col, ...; the last is the kernel's own line
        locs = [x for x in ch if re.search(r":\d+:\d+$", x)]
        line = int(locs[-1].rsplit(":", 2)[1]) if locs else 0
        r = region(line) if line else regs[0][1]
        if tile and r == "H: window fetch" and any(slow_lo <= int(x.rsplit(":", 2)[1]) <= slow_hi for x in locs[:-1]):
            r = "H: window fetch, clamped (picture border)"
        c = classify(mn)
        counts[r][c] += 1
        mnems[c][mn] += 1
    note = subprocess.run([LLVM + "llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    meta = ""
    for entry in re.split(r"\n  - (?=\.)", note):  # one list entry per kernel
        if re.search(r"\.name:\s+" + re.escape(name) + r"\s", entry):
            got = []
            for key in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
                m = re.search(r"\." + key + r":\s+(\d+)", entry)
                if m:
                    got.append("%s=%s" % (key, m.group(1)))
            meta = ", ".join(got)
    print("kernel: %s" % name)
    if meta:
        print("resources (code object metadata): %s" % meta)
    print("%d instructions; static counts (every path once, unrolled copies counted as often as they were emitted)\n" % len(insts))
    classes = [c for c, _ in CLASSES if any(counts[r][c] for r in counts)]
    first_scalar = classes.index("LDS") if "LDS" in classes else len(classes)
    valu = classes[:first_scalar]
    w = max(len(c) for c in classes) + 2
    print("%-*s" % (w, "class") + "".join("%7s" % ("R%d" % i) for i in range(len(order))) + "%7s" % "all")
    for c in classes:
        row = [counts[r][c] for r in order]
        print("%-*s" % (w, c) + "".join("%7d" % v for v in row) + "%7d" % sum(row))
    tot_valu = [sum(counts[r][c] for c in valu) for r in order]
    print("%-*s" % (w, "vector ALU, all classes") + "".join("%7d" % v for v in tot_valu) + "%7d" % sum(tot_valu))
    print()
    for i, nm in enumerate(order):
        print("R%d = %s" % (i, nm))
    if tile and "Li64ELi512" in name:
        # What one wave of a 64x64 region executes (interior region, fractional vectors both ways).  The two lists are
        # unrolled copies (a region predicted from one list runs one of them: the per-list regions count half), the
        # V loop's two rounds are copies as well (chroma work only exists in the second: half of the chroma copy), and
        # of the 8 waves 5 own luma H items (the fifth only half a wave of them), 3 own chroma H items, 4 own chroma V items.
        V = dict(zip(order, tot_valu))
        D = dict(zip(order, [counts[r]["dot product"] for r in order]))

        def wave(T, lists, h_luma, v_chroma):
            f = lists / 2.0
            n = T["entry, region description"] + T["H: filter set-up, item decode"] + T["barrier, V set-up (rounding, weights)"] + T["V: luma round, clip, store"]
            n += f * (T["H: window fetch"] + T["H: pack and store to LDS"] + T["V: luma"])
            n += f * (T["H: luma 8-tap"] if h_luma else T["H: chroma 4-tap"])
            if v_chroma:
                n += 0.5 * (0.5 + 0.5 * f) * 2 * T["V: chroma"] * 0.5  # one of two copies; its filter half scales with the lists
            return n

        print("\nmodel of the executed vector-ALU instructions per wave (see the comment in tools/isa_census.py):")
        for lists in (2, 1):
            tot = {"valu": 0.0, "dot": 0.0}
            for key, T in (("valu", V), ("dot", D)):
                roles = [wave(T, lists, True, True)] * 4 + [wave(T, lists, True, False)] + [wave(T, lists, False, False)] * 3
                tot[key] = sum(roles) / 8
                if key == "valu":
                    print("  %d list%s: waves 0-3 %.0f, wave 4 %.0f, waves 5-7 %.0f, mean %.0f" % (lists, "s" if lists > 1 else "", roles[0], roles[4], roles[5], tot[key]))
            print("           of which dot products: mean %.0f (%.0f %%)" % (tot["dot"], 100 * tot["dot"] / tot["valu"]))
    print("\nmnemonics per class:")
    for c in classes:
        print("  %s: %s" % (c, ", ".join("%s %d" % kv for kv in mnems[c].most_common())))


if __name__ == "__main__":
    main()
