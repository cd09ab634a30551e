#!/usr/bin/env python3
"""Writes the NUMBER tables of profiles/README.md and DESIGN.md (section 7; rounds before 5: profiles/HISTORY.md) from the files under profiles/ -- nothing in those two
blocks is typed by hand (three documents quoted three sets of numbers for one profile file).
usage: tools/gen_results.py [tag]            rewrite the blocks between `<!-- results:<tag>:begin -->` / `<!-- results:<tag>:end -->`
       tools/gen_results.py [tag] --check    exit 1 if a block differs from what the files give (tests/test_docs_numbers.py)"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _load(name):
    try:
        with open(os.path.join(P, name)) as fh:
            return json.loads(fh.read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None


def _text(name):
    try:
        with open(os.path.join(P, name)) as fh:
            return fh.read()
    except OSError:
        return None


def _stage_of(kernel):
    k = kernel
    if k.startswith("hrtail_") or k.startswith("lk5_") or k.startswith("lk_conv_kernel") or k.startswith("to_nhwc16") or k.startswith("pack_kernel"):
        return "HR stage's own kernels (5x5 conv, border terms, collapse / expand, layout of the image gradient)"
    if k.startswith("conv_trunk_kernel"):
        return "conv_trunk: the trunk's 33 convolutions per direction in one image-stationary launch (+ the long skip's add in the backward one)"
    if k.startswith("conv_pair_kernel"):
        return "conv_pair: one launch per ResBlock / RCAB conv pair and direction"
    if k.startswith("ca_") or k.startswith("rowsum_group"):
        return "channel attention (stand-alone launches)"
    if k.startswith("conv_wgrad_ws_group") or k.startswith("wgrad_finalize_group") or k.startswith("upload_kernel"):
        return "grouped 3x3 weight gradients (body + first upsampler stage)"
    if k.startswith("conv_ws_kernel") or k.startswith("conv_ks_kernel"):
        return "conv_ws / conv_ks forward / data-gradient launches (body, first upsampler stage; layer by layer: the HR stage too)"
    if k.startswith("l1_") or "FillFunctor" in k:
        return "L1 loss"
    if k.startswith("adam_") or "copyBuffer" in k:
        return "Adam (one launch) + its table"
    return "head conv, packing, layout, the long skip's gradient add"


def step_anatomy(name):
    t = _text(name)
    if not t:
        return None
    agg = collections.OrderedDict()
    total = None
    for line in t.splitlines():
        m = re.match(r"^(.*?)\s+(x?\d+)\s+([\d.]+)(\s+\(.*\))?$", line)
        if line.startswith("sum of kernel durations"):
            total = float(line.split()[4])
            continue
        if not m:
            continue
        kern, cnt, us = m.group(1).strip(), m.group(2), float(m.group(3))
        n = int(cnt[1:]) if cnt.startswith("x") else 1
        st = _stage_of(kern)
        a = agg.setdefault(st, [0, 0.0])
        a[0] += n
        a[1] += us
    return agg, total


def block(tag, short=False):
    """short: DESIGN.md's block (the lines, the tables of the default step and of all models); everything else -- rocprofv3 flavours, PMC
    passes, stamps, A/B files -- only in profiles/README.md's block."""
    out = []
    d = _load(f"{tag}_bench_default.json")
    if d:
        c, r = d["config"], d.get("roofline", {})
        out.append(f"**Default line** (`profiles/{tag}_bench_default.json`, `python bench.py`): **{d['value']:,.0f} LR patches/s**, {d['ms_per_step']:.3f} ms per step "
                   f"(sustained {d.get('sustained_value', 0):,.0f}); `model_mfma_frac` {d['model_mfma_frac']:.4f} on the reference graph's {c['gflop_per_patch']['reference_graph']} GFLOP per patch, "
                   f"{d['model_mfma_frac_executed']:.4f} on the {c['gflop_per_patch']['executed']} GFLOP the path executes; HR stage: {c['hr_stage'].split(':')[0]}.")
        lw = d.get("layerwise_hr_stage")
        if lw and "value" in lw:
            out.append(f"Same process, same step with the HR stage layer by layer (`layerwise_hr_stage`): {lw['value']:,.0f} patches/s, {lw['ms_per_step']:.3f} ms per step "
                       f"(`model_mfma_frac` {lw['model_mfma_frac']:.4f}): the collapsed form is {d['value'] / lw['value']:.2f}x.")
        if r and "isolated" in r:                       # round 5: ONE fraction, measured in the step; the isolated flavours in a sub-object
            iso, ins = r["isolated"], r.get("in_step")
            out.append("")
            out.append(f"`roofline` ({r['kernel']}): **{r['us_per_launch']} us per launch = {r['achieved']} TFLOP/s = {r['frac']:.4f} of 2.5 PFLOP/s** -- {r.get('where')}; "
                       f"HBM-side traffic per launch {('%.1f MB' % (r['traffic'] / 1e6)) if r.get('traffic') else 'null'} against {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic.")
            if ins:
                g = ins["graph_us"]
                out.append(f"`in_step`: {ins['convs']} convolutions ({ins['convs_per_launch']} per launch); forward {ins['fwd_us_per_conv']} us, data gradient {ins['dgrad_us_per_conv']} us, "
                           f"grouped weight gradient {ins['wgrad_us_per_layer']} us per layer; with the weight gradients the trunk runs at {ins['frac_with_wgrad']} of the peak "
                           f"(graphs: packing {g['pack']} us, forward {g['fwd']}, + data gradients {g['fwd_bwd']}, + weight gradients {g['fwd_bwd_wgrad']}).")
            v, vb = iso["variants_us"], iso.get("variants_us_burst", {})
            out.append(f"`isolated` (each flavour alone on the chip, re-reading its own buffers): {iso['quoted_flavour']} {iso['us_per_launch']} us = {iso['frac']} (burst {iso.get('frac_burst')}); "
                       f"step-weighted over a block's launches {iso.get('step_weighted_frac')} (burst {iso.get('step_weighted_frac_burst')}).")
            out.append("")
            out.append("| flavour (HIP events, isolated) | sustained us | burst us | launches per block |")
            out.append("|---|---|---|---|")
            for k in v:
                out.append(f"| {k} | {v[k]} | {vb.get(k)} | {iso.get('launches_per_block', {}).get(k)} |")
        elif r and "variants_us" in r:
            v, vb = r["variants_us"], r.get("variants_us_burst", {})
            out.append("")
            out.append(f"`roofline` ({r['kernel']}): sustained {r['us_per_launch']} us per launch = {r['achieved']} TFLOP/s = **{r['frac']:.4f}** of 2.5 PFLOP/s "
                       f"(burst {r.get('frac_burst')}); HBM-side traffic per launch {('%.1f MB' % (r['traffic'] / 1e6)) if r.get('traffic') else 'null'} "
                       f"against {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic; step-weighted over a ResBlock's launches **{r.get('step_weighted_frac')}** (burst {r.get('step_weighted_frac_burst')}).")
            out.append("")
            out.append("| flavour (HIP events, this line) | sustained us | burst us | launches per ResBlock |")
            out.append("|---|---|---|---|")
            for k in v:
                out.append(f"| {k} | {v[k]} | {vb.get(k)} | {r.get('launches_per_block', {}).get(k)} |")
        cb = d.get("cpu_baseline")
        if cb and "value" in cb:
            out.append("")
            if cb.get("cpu_model"):
                out.append(f"`cpu_baseline` host: {cb['cpu_model']}, {cb.get('cores_available')} cores available, {cb['cores']} threads used.")
            par = cb.get("parity", {})
            bw = par.get("bench_weights", par)
            out.append(f"`cpu_baseline`: {cb['value']} patches/s ({cb['sample']}); parity of the build on the bench's own weights: PSNR(build, oracle) {bw.get('psnr_build_vs_oracle_db')} dB, max |err| {bw.get('max_abs_err')}.")
            ac = cb.get("all_cores")
            if ac and "value" in ac:
                out.append(f"`cpu_baseline.all_cores`: {ac['value']} patches/s ({ac['sample']}).")
            elif ac:
                out.append(f"`cpu_baseline.all_cores`: {ac.get('note')}.")
            tn = par.get("trained_net")
            if tn and "delta_psnr_db" in tn:
                d_, w_ = tn["delta_psnr_db"], tn.get("worst_image_delta_db", {})
                out.append(f"`cpu_baseline.parity.trained_net` ({tn['net']}; PSNR of the reference path {tn['psnr_reference_path_db']} dB, criterion {tn['criterion_db']} dB): "
                           + "; ".join(f"{k} {d_[k]:+.4f} dB (worst image {w_.get(k, 0):+.4f})" for k in d_) + ".")
        oc = d.get("other_configs")
        if oc:
            out.append("")
            out.append("| `other_configs` (batch 16, training step as one hipGraph) | patches/s | ms / step | model_mfma_frac (reference graph / executed) | dominant kernel frac (in the step where the model has a trunk) | isolated frac / step-weighted | in-step PMC traffic / algorithmic bytes per launch (kernel) |")
            out.append("|---|---|---|---|---|---|---|")
            for e in oc:
                if "value" not in e:
                    out.append(f"| {e.get('model')} {e.get('dtype', '')} | error: {e.get('error')} | | | | | |")
                    continue
                rr = e.get("roofline") or {}
                iso = rr.get("isolated") or {}
                out.append(f"| {e['model']}{' fp16' if e.get('dtype') == 'f16' else ''} | {e['value']:,.0f} | {e['ms_per_step']} | {e['model_mfma_frac']} / {e.get('model_mfma_frac_executed')} | "
                           f"{rr.get('frac', '')} {('(' + rr.get('unit', '') + (', in step' if 'in_step' in rr else ', isolated') + ')') if rr else ''} | "
                           f"{iso.get('frac', rr.get('frac', ''))} / {iso.get('step_weighted_frac', rr.get('step_weighted_frac', ''))} | "
                           + (f"{rr['traffic'] / 1e6:.1f} MB / {rr['algorithmic_bytes_per_launch'] / 1e6:.1f} MB ({rr.get('traffic_kernel')})" if rr.get("traffic") and rr.get("algorithmic_bytes_per_launch") else "") + " |")
    for nm, what in ((f"{tag}_bench_default_f16.json", "fp16 (`--dtype f16`, device-resident dynamic loss scaling, the step still ONE hipGraph)"),
                     (f"{tag}_bench_inference.json", "forward only (`--inference`)")):
        e = _load(nm)
        if e:
            extra = ""
            ls = e.get("config", {}).get("loss_scale")
            if ls:
                extra = f"; loss scale {ls['scale']:g}, {ls['skipped_steps']} skipped steps"
            out.append("")
            out.append(f"{what}: **{e['value']:,.0f} patches/s**, {e['ms_per_step']:.3f} ms per step (`profiles/{nm}`{extra}).")
    t = None if short else _text(f"{tag}_variants_n256.txt")
    if t:
        out.append("")
        out.append(f"The same flavours under rocprofv3 (`profiles/{tag}_variants_n256.txt`: isolated launches replayed for 1.5 s, second half of the dispatches averaged):")
        out.append("")
        out.append("```")
        out += [l[:170] for l in t.strip().splitlines()[2:]]
        out.append("```")
    for nm, title in ((f"{tag}_step_edsr_baseline_b256.txt", "Where the default step goes (one step of the rocprofv3 kernel trace in dispatch order, grouped by stage)"),
                      (f"{tag}_step_edsr_baseline_b256_layerwise.txt", "The same with the HR stage layer by layer (`SRK_DEBUG=1 SRK_NO_HR_COLLAPSE=1`)"),
                      (f"{tag}_step_edsr_baseline_b256_per_layer.txt", "The same with the trunk as one launch per convolution (`SRK_DEBUG=1 SRK_NO_TRUNK=1`)"),
                      (f"{tag}_step_edsr_baseline_b16.txt", "EDSR-baseline at the reference's batch of 16"), (f"{tag}_step_rcan_b16.txt", "RCAN at batch 16"),
                      (f"{tag}_step_rcan_b64.txt", "RCAN at batch 64"), (f"{tag}_step_rcan_b256.txt", "RCAN at batch 256")):
        sa = None if (short and ("layerwise" in nm or "rcan_b64" in nm)) else step_anatomy(nm)
        if sa:
            agg, total = sa
            out.append("")
            out.append(f"{title} (`profiles/{nm}`; sum of kernel durations {total:,.0f} us):")
            out.append("")
            out.append("| stage | launches | us | share |")
            out.append("|---|---|---|---|")
            for k, (n, us) in agg.items():
                out.append(f"| {k} | {n} | {us:,.0f} | {100 * us / total:.1f} % |")
    t = _text(f"{tag}_sweep.txt")
    if t:
        out.append("")
        out.append(f"All models, one box (`profiles/{tag}_sweep.txt`):")
        out.append("")
        out.append("```")
        out += t.strip().splitlines()
        out.append("```")
    if short:
        out.append("")
        out.append(f"rocprofv3 timings of the isolated flavours, PMC passes, in-kernel stamps and the same-box A/B files: `profiles/README.md` (generated from the same files).")
        return "\n".join(out) + "\n"
    pm = sorted(f for f in os.listdir(P) if f.startswith(tag + "_") and f.endswith("_pmc.txt")) if os.path.isdir(P) else []
    if pm:
        out.append("")
        out.append("PMC passes (`tools/pmc_kernel.sh`: one counter group per rocprofv3 run, FETCH_SIZE and WRITE_SIZE separately, FETCH_SIZE x 2 on gfx950):")
        out.append("")
        out.append("| file | kernel | MFMA pipe busy | HBM-side traffic per launch | LDS bank conflicts |")
        out.append("|---|---|---|---|---|")
        for f in pm:
            t = _text(f)
            # one section per kernel name; quote the one with the most dispatches (a variant's file also holds the single set-up launch)
            secs, cur = [], None
            for line in t.splitlines():
                if line.strip() and not line.startswith((" ", "#")):
                    cur = dict(kern=line.strip().replace("void ", "").replace("(anonymous namespace)::", "")[:60], n=0, busy="", hbm="", conf="")
                    secs.append(cur)
                    continue
                if cur is None:
                    continue
                m = re.search(r"\sn=\s*(\d+) mean=", line)
                if m:
                    cur["n"] = max(cur["n"], int(m.group(1)))
                m = re.search(r"MFMA pipe busy ([\d.]+)", line)
                if m:
                    cur["busy"] = m.group(1)
                m = re.search(r"HBM-side traffic per launch: ([\d.]+ MB)", line)
                if m:
                    cur["hbm"] = m.group(1)
                m = re.search(r"SQ_LDS_BANK_CONFLICT\s+n=\s*\d+ mean=\s*([\d.]+)", line)
                if m:
                    cur["conf"] = f"{float(m.group(1)):,.0f}"
            if not secs:
                continue
            best = max(secs, key=lambda c: c["n"])
            kern, busy, hbm, conf = best["kern"], best["busy"], best["hbm"], best["conf"]
            out.append(f"| `{f}` | {kern} | {busy} | {hbm} | {conf} |")
    for nm, title in ((f"{tag}_ab_pair.txt", "conv_pair, same box, round 4's library against this tree (`tools/microbench_pair.py 16`: a dependent chain of 32 ResBlock pairs)"),
                      (f"{tag}_ab_pw_b16.txt", "WDSR-B at batch 16, same box, round 5's two changes switched off and on (`tools/ab_pw.sh 16`: patches/s, ms per step)"),
                      (f"{tag}_stamps.txt", "In-kernel `s_memtime` stamps of workgroup 0 (diagnostics build, `tools/stamp_*.py`; ticks ~ cycles)"),
                      (f"{tag}_hrtail_microbench.txt", "The HR stage alone, forward + backward, eager launches (`tools/microbench_hrtail.py`)"),
                      (f"{tag}_ab_ddp.txt", "Single process against a forced 1-rank RCCL group, same box (`tools/ab_ddp.sh`: value, ms per step, sustained value, graph form, gradient sync)"),
                      (f"{tag}_ab_trunk.txt", "The trunk as one image-stationary launch per direction against one launch per convolution, same box (`tools/microbench_trunk.py`; `SRK_NO_TRUNK`)"),
                      (f"{tag}_exact_relu.txt", "The NaN-preserving-ReLU build against the default build, same box (`tools/ab_exact_relu.sh`: patches/s, ms per step)"),
                      (f"{tag}_experiments.txt", "Round 6's same-box experiments on the way to the trunk launch")):
        t = _text(nm)
        if t:
            out.append("")
            out.append(f"{title}, `profiles/{nm}`:")
            out.append("")
            out.append("```")
            out += t.strip().splitlines()
            out.append("```")
    return "\n".join(out) + "\n"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r6"
    check = "--check" in sys.argv
    b, e = f"<!-- results:{tag}:begin -->\n", f"<!-- results:{tag}:end -->"
    bad = 0
    first = os.path.join(ROOT, "DESIGN.md") if tag == "r6" else os.path.join(P, "HISTORY.md")      # earlier rounds' results live in profiles/HISTORY.md
    for path in (first, os.path.join(P, "README.md")):
        body = block(tag, short=(path.endswith("DESIGN.md")))
        s = open(path).read()
        if b not in s or e not in s:
            print(f"{path}: markers for {tag} not found")
            bad += 1
            continue
        i, j = s.index(b) + len(b), s.index(e)
        if check:
            if s[i:j] != body:
                print(f"{path}: the generated block is out of date (run tools/gen_results.py {tag})")
                bad += 1
        else:
            open(path, "w").write(s[:i] + body + s[j:])
            print(f"{path}: block {tag} written ({len(body.splitlines())} lines)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
