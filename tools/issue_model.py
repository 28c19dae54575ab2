#!/usr/bin/env python3
"""Issue model of pwn_trace_kernel<false,false,false,true> (no counters, 3 lanes, unit order, inline sphere records): what its instruction stream costs the SIMDs, region by region.

    static    the kernel's ISA (hipcc -S with line tables, the Makefile's flags), every instruction attributed to a REGION
              of the source through its .loc chain (the `//@R name` comments of trace_kernel.hip / trace_walk.inc mark
              the regions; an instruction inlined from dev_math.h etc. belongs to the region of its call site) and
              classed by opcode: full-rate VALU, half-rate VALU, quarter-rate VALU (rcp / sqrt / div), scalar ALU,
              branch, LDS, vector memory, other (waitcnt, nop)
    dynamic   how often a wave64 runs each region with at least one lane: the counting variant's counters
              (pwn_stats.wave_steps, wave_paths, regions), taken on the GPU by tools/region_counts.py
              -> profiles/r5_region_counts.json
    costs     ns of SIMD issue per wave-instruction, tools/ubench/valu_rate.hip (profiles/r3_valu_rate.txt): full-rate
              1 / 0.91, half-rate 1 / 0.545, quarter-rate 1 / 0.29 at 5 waves per SIMD with one opcode; 1 / 1.00, 1 / 0.57,
              1 / 0.293 saturated.  Scalar and branch instructions are counted, not priced: they issue beside the VALU
              instructions of other waves (the table shows what they would add if they did not)

    python3 tools/issue_model.py [--counts profiles/r5_region_counts.json] [--out profiles/r5_issue_model]

Prints the per-region table for the headline scene, the instruction totals against the PMC counters of the same
launch (profiles/pmc_latest.csv: the check that the COUNTS are right, independent of any timing), and predicted
against measured launch times; writes <out>.txt and <out>.json (bench.py reads the latter for roofline.issue_frac)."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pwnfps_amd", "csrc")
KERNEL = "_Z16pwn_trace_kernelILb0ELb0ELb0ELb1EEv16pwn_trace_params"

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_bitop3_b32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32",
        "v_sub_co_u32", "v_subb_co_u32", "v_xnor_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
           "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
# ns of SIMD issue per wave-instruction (profiles/r3_valu_rate.txt): at 5 waves per SIMD -- what this kernel runs with -- the
# microbenchmark's streams of ONE opcode reach 0.91 / 0.545 / 0.29 instructions per ns (full / half / quarter rate); at 8 waves
# 1.00 / 0.57 / 0.293, the pipes' saturated rates, which a MIXED stream of five waves also reaches (mix_vs: 0.96 at 5 waves)
COST = {"full": 1.0 / 0.91, "half": 1.0 / 0.545, "quarter": 1.0 / 0.29}
COST_SAT = {"full": 1.0 / 1.00, "half": 1.0 / 0.57, "quarter": 1.0 / 0.293}


def opclass(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_"):
        if op.endswith("_dpp") or op.endswith("_sdwa"):
            return "half"
        if base in QUARTER:
            return "quarter"
        if base in FULL:
            return "full"
        return "half"                                      # compares, selects, shifts, converts, integer multiplies, min / max / med3, packed, f64 ...
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_setpc_b64", "s_swappc_b64"):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op in ("s_endpgm", "s_barrier", "s_sleep", "s_setprio"):
        return "other"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime") or op.startswith("s_dcache"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def slow_ranges():
    """line ranges of dev_math.h between //@SLOW and //@FAST: the general paths behind wave-uniform branches that ordinary
    frames never take (non-finite rays, |angle| >= 120, fog beyond e^-88 ...)"""
    out, start = [], None
    for i, line in enumerate(open(os.path.join(CSRC, "dev_math.h")), 1):
        if "//@SLOW" in line:
            start = i
        elif "//@FAST" in line and start is not None:
            out.append((start, i))
            start = None
    return out


def region_maps():
    """file -> sorted [(line, region)] from the //@R markers"""
    out = {}
    for name in ("trace_kernel.hip", "trace_walk.inc", "trace_sphere.inc"):
        marks = []
        for i, line in enumerate(open(os.path.join(CSRC, name)), 1):
            m = re.search(r"//@R (\w+)", line)
            if m:
                marks.append((i, m.group(1)))
        out[name] = marks
    return out


def region_of(marks, name, line):
    r = None
    for ln, reg in marks[name]:
        if ln <= line:
            r = reg
        else:
            break
    return r


def build_asm(path):
    flags = subprocess.check_output(["make", "-s", "-C", CSRC, "--no-print-directory", "-pn"], text=True, stderr=subprocess.DEVNULL)
    fp = re.search(r"^FPFLAGS = (.*)$", flags, re.M).group(1)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + fp.split() + [
        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-DPWN_MIN_WAVES=5", "-mllvm", "-disable-lsr"] + os.environ.get("PWN_ISA_EXTRA", "").split() + ["-gline-tables-only", "-S", "--cuda-device-only",
        "-o", path, os.path.join(CSRC, "trace_kernel.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def parse(path, marks):
    """[(block, region, class, opcode)] of the kernel.  An instruction's region comes from its .loc chain; the compiler's
    own instructions carry none (exec-mask restores and selects in the join block of an `if`): inside a block they take the
    region of the located instruction that follows them (else the one before); a block without any located instruction
    is a join block and takes the region of the first branch that targets it -- the head of the `if` it closes."""
    slow = slow_ranges()
    on = False
    blocks = []                                   # [label, [(region or None, class, op, branch target or None)]]
    loc_region = None
    for line in open(path):
        if line.startswith(KERNEL + ":"):
            on = True
            blocks.append(["entry", []])
            continue
        if not on:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %bb\.(\d+):", line)
        if m:
            blocks.append([m.group(1), []])
            loc_region = None                       # (a .loc does not carry over a label)
            continue
        t = line.strip()
        if t.startswith(".loc"):
            # innermost frame first; take the first one that lies in the kernel's own sources
            frames = re.findall(r"([\w./+-]+):(\d+):\d+", t.split(";", 1)[1] if ";" in t else "")
            loc_region = None
            slow_hit = any(os.path.basename(f) == "dev_math.h" and any(a <= int(ln) <= b for a, b in slow) for f, ln in frames)
            for f, ln in frames:
                base = os.path.basename(f)
                if base in marks and int(ln) > 0:
                    loc_region = region_of(marks, base, int(ln))
                    break
            if loc_region and slow_hit:
                loc_region += "~slow"
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        tgt = t.split()[-1] if op.startswith(("s_cbranch", "s_branch")) else None
        blocks[-1][1].append([loc_region, opclass(op), op, tgt])
    # inside a block: backward fill from the next located instruction, then forward fill
    for label, ins in blocks:
        nxt = None
        for i in range(len(ins) - 1, -1, -1):
            if ins[i][0] is None:
                ins[i][0] = nxt
            else:
                nxt = ins[i][0]
        prev = None
        for it in ins:
            if it[0] is None:
                it[0] = prev
            else:
                prev = it[0]
    # join blocks: the region of the first branch that targets them; blocks that are only fallen into: the block before
    src = {}
    for label, ins in blocks:
        for it in ins:
            if it[3] and it[3] not in src and it[0] is not None:
                src[it[3]] = it[0]
    prev_region = "k_prologue"
    out = []
    for label, ins in blocks:
        if ins and ins[0][0] is None:
            r = src.get(label, prev_region)
            for it in ins:
                it[0] = r
        for it in ins:
            out.append((label, it[0], it[1], it[2]))
            prev_region = it[0]
    return out


COUNT_OF = {        # region -> key of the counts
    "k_prologue": "waves", "k_epilogue": "waves", "k_unit": "units", "k_help": "help",
    "p_setup": "segs", "p_setup_slow": "setup_slow", "p_walk_ctl": "wave_steps", "p_post": "segs", "p_exhausted": "exhausted_w",
    "p_wall": "wall", "p_sphere": "sphere", "p_floor": "floor", "p_sphrefl": "sphrefl", "p_jitter": "jitter",
    "p_comp": "units", "p_comp1": "comp1", "p_comp1_fog": "comp1_fog", "p_comp2": "comp2", "p_comp2_fog": "comp2_fog",
    "w_head": "wave_steps", "w_sphlist": "wp0", "w_sphtest": "sphtest", "w_sphhit": "wp7", "w_sphupd": "sphupd",
    "w_room": "wp1", "w_fog": "wp2", "w_height": "wp3", "w_height_r2": "hc_r2", "w_height_out": "hc_out", "w_else": "else", "w_ramp": "wp4",
    "w_portal": "wp5", "w_portal_wall": "portal_wall", "w_portal_go": "portal_go", "w_portal_odd": "portal_odd", "w_portal_rot2": "portal_rot2",
    "w_solid": "wp6", "k_unit_half": "unit_half",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counts", default=os.path.join(ROOT, "profiles", "r4_region_counts.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r5_issue_model"))
    ap.add_argument("--asm", default=None, help="an existing .s (else compiled here)")
    args = ap.parse_args()

    marks = region_maps()
    asm = args.asm
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(), "trace_kernel.s")
        build_asm(asm)
    ins = parse(asm, marks)
    per = collections.defaultdict(collections.Counter)           # region -> class -> n
    blocks = collections.defaultdict(set)
    for b, reg, cl, op in ins:
        per[reg][cl] += 1
        blocks[reg].add(b)
    lines = []
    P = lines.append
    P("# issue model of pwn_trace_kernel<false,false,false,true> (tools/issue_model.py); static part: %d instructions in %d regions" % (len(ins), len(per)))
    P("# costs per wave-instruction and SIMD at 5 waves / SIMD (profiles/r3_valu_rate.txt): full-rate VALU %.2f ns, half-rate %.2f ns, quarter-rate %.2f ns"
      % (COST["full"], COST["half"], COST["quarter"]))
    P("")
    P("%-14s %6s | %5s %5s %5s | %5s %5s %4s %4s %5s" % ("region", "blocks", "full", "half", "quart", "salu", "br", "lds", "vmem", "other"))
    order = list(COUNT_OF)
    for reg in sorted(per, key=lambda r: (order.index(r.split("~")[0]) if r.split("~")[0] in order else 99, r)):
        c = per[reg]
        P("%-14s %6d | %5d %5d %5d | %5d %5d %4d %4d %5d" % (reg, len(blocks[reg]), c["full"], c["half"], c["quarter"], c["salu"] + c["smem"], c["branch"],
                                                        c["lds"], c["vmem"], c["other"]))
    missing = [r for r in per if r not in COUNT_OF and not r.endswith("~slow")]
    if missing:
        P("# regions without a counter: %s" % missing)

    if not os.path.exists(args.counts):
        print("\n".join(lines))
        print("\n(no %s: run tools/region_counts.py on the GPU for the dynamic part)" % args.counts)
        return
    scenes = json.load(open(args.counts))["scenes"]
    # the instruction counters of the 4K level.txt launch: the committed PMC summary of the same build
    try:
        rows = [ln.rstrip("\n").rsplit(",", 3) for ln in open(os.path.join(ROOT, "profiles", "pmc_latest.csv"))]
        v = {r[1]: float(r[3]) for r in rows if len(r) == 4 and "pwn_trace_kernel<false, false, false, true>" in r[0]}
        for sc in scenes:
            if (sc["level"], sc["w"], sc["h"]) == ("pwnfps_level", 3840, 2160):
                sc["pmc"] = {k: v[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS")}
    except (OSError, KeyError, ValueError):
        pass

    def totals(cnt):
        t = collections.Counter()
        by_region = {}
        for reg, c in per.items():
            n = cnt.get(COUNT_OF.get(reg, ""), 0)
            valu_ns = n * (c["full"] * COST["full"] + c["half"] * COST["half"] + c["quarter"] * COST["quarter"])
            sat_ns = n * (c["full"] * COST_SAT["full"] + c["half"] * COST_SAT["half"] + c["quarter"] * COST_SAT["quarter"])
            sc = n * (c["salu"] + c["smem"] + c["branch"])
            by_region[reg] = (n, n * (c["full"] + c["half"] + c["quarter"]), sc, valu_ns, n * c["half"])
            t["valu"] += n * (c["full"] + c["half"] + c["quarter"])
            t["half"] += n * c["half"]
            t["salu"] += n * (c["salu"] + c["smem"])
            t["branch"] += n * c["branch"]
            t["lds"] += n * c["lds"]
            t["valu_ns"] += valu_ns
            t["sat_ns"] += sat_ns
            t["scalar_n"] += sc
        return t, by_region

    P("")
    P("# Reading: `VALU issue` is what the kernel's VALU instructions alone cost the 1024 SIMDs -- entries x instructions x issue cost, nothing")
    P("# overlapped, nothing else counted: the floor of the launch.  `span` is the launch from its first wave's start to its last wave's end")
    P("# (wave stamps, PWN_OPT_WAVE_LOG), `busy` the mean wave's lifetime.  Where VALU issue at the saturated rates comes to ~100 % of the")
    P("# span, the launch is AT its VALU-issue floor and its scalar third rides along in the shadow of other waves' VALU instructions")
    P("# (additive at the ~1.0 ns they cost beside VALU work in the microbenchmark they would add a third).  Launches whose span is well")
    P("# above the floor are bound by their tail (few units per wave, or single waves walking mirror halls to the step limit): section 6.")
    out_cases = []
    for sc in scenes:
        t, by_region = totals(sc["counts"])
        simds = sc.get("simds", 1024)
        valu_ms = t["valu_ns"] / simds * 1e-6
        sat_ms = t["sat_ns"] / simds * 1e-6
        res = sc.get("residency", 1.0)
        span = sc.get("span_ms", sc["trace_ms"])
        busy = span * res
        P("")
        P("## %s %dx%d: launch span %.4f ms (wave stamps; %.4f ms between HIP events around a blocking frame), mean wave residency %.3f -> busy %.4f ms" % (
            sc["level"], sc["w"], sc["h"], span, sc["trace_ms"], res, busy))
        P("   wave-instructions by the model: VALU %.4g (half-rate %.4g), SALU %.4g, branch %.4g, LDS %.4g" % (t["valu"], t["half"], t["salu"], t["branch"], t["lds"]))
        if sc.get("pmc"):
            pm = sc["pmc"]
            P("   PMC of the same launch:         VALU %.4g, SALU %.4g, branch %.4g, LDS %.4g   (model / PMC: %.3f %.3f %.3f %.3f)" % (
                pm["SQ_INSTS_VALU"], pm["SQ_INSTS_SALU"], pm["SQ_INSTS_BRANCH"], pm["SQ_INSTS_LDS"], t["valu"] / pm["SQ_INSTS_VALU"],
                t["salu"] / pm["SQ_INSTS_SALU"], t["branch"] / pm["SQ_INSTS_BRANCH"], t["lds"] / pm["SQ_INSTS_LDS"]))
        P("   VALU issue: %.4f ms at the saturated rates (%.4f at the 5-wave single-opcode rates) = %.3f of the span (predicted launch %+.1f %% against it), %.3f of busy;" % (
            sat_ms, valu_ms, sat_ms / span, (sat_ms / span - 1) * 100, sat_ms / busy))
        P("   scalar + branch instructions: %.4g (x 1.0 ns = %.4f ms if they were additive)" % (t["scalar_n"], t["scalar_n"] / simds * 1e-6))
        P("   %-14s %10s %12s %10s %10s %7s" % ("region", "entries", "VALU instr", "half-rate", "VALU ms", "share"))
        for reg in sorted(by_region, key=lambda r: -by_region[r][3]):
            n, nv, nsc, vns, nh = by_region[reg]
            if n:
                P("   %-14s %10d %12d %10d %10.4f %6.1f%%" % (reg, n, nv, nh, vns / simds * 1e-6, 100 * vns / max(t["valu_ns"], 1)))
        out_cases.append({"level": sc["level"], "w": sc["w"], "h": sc["h"], "valu_issue_ms": round(sat_ms, 4), "valu_issue_ms_5_wave_rates": round(valu_ms, 4),
                          "span_ms": span, "residency": res, "valu": int(t["valu"]), "valu_half_rate": int(t["half"]), "scalar_and_branch": int(t["scalar_n"])})
    txt = "\n".join(lines) + "\n"
    print(txt)
    with open(args.out + ".txt", "w") as f:
        f.write(txt)
    with open(args.out + ".json", "w") as f:
        json.dump({"costs_ns_5_waves": COST, "costs_ns_saturated": COST_SAT, "cases": out_cases}, f, indent=1)


if __name__ == "__main__":
    main()
