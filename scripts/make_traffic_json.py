"""Turns the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_passes.sh into profiles/<name>.json:
per-kernel HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled; both are in KiB).
usage: python scripts/make_traffic_json.py <pmc_dir> <out.json> <workload> [<valu_out.json>]
With the fourth argument also the VALU wave-instructions per launch of every entry point (SQ_INSTS_VALU of
pass 1) -- what bench.py reports as roofline.valu."""
import json
import re
import sys

pmc_dir, out, workload = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for name in ("pass3.txt", "pass4.txt"):
    for line in open(f"{pmc_dir}/{name}"):
        m = re.match(r"(.{60}) (\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m:
            k = re.sub(r"\(.*", "", m.group(1).strip()).replace("void ", "")
            vals.setdefault(k, {})[m.group(2)] = float(m.group(3))
ENTRY_UNCUT = {
    "gs_render_tiles_backward_slab": ["gs::k_render_bwd<float, 1>", "gs::k_tile_order"],
    "gs_render_tiles_prefix": ["gs::k_render_fwd<float, 1>", "gs::k_render_fwd_flagged", "gs::k_render_fwd_flagged<false>",
                               "gs::k_tile_sort_flagged<8192>", "gs::k_tile_sort_flagged<4096>"],
    "gs_preprocess_forward": ["gs::k_preprocess<16, false>", "gs::k_cull_count", "gs::k_scan_counts"],
    "gs_preprocess_backward": ["gs::k_preprocess_bwd<16>", "gs::k_preprocess_bwd<16, false>"],
    "gs_tile_count": ["gs::k_bin_count", "gs::k_bin_colscan", "gs::k_bin_colscan<64>", "gs::k_scan_tiles"],
    "gs_tile_emit_sort": ["gs::k_bin_emit", "gs::k_tile_sort<true>", "gs::k_tile_sort_big<true>"],
}
# frames binned with the depth cut (csrc/binning.hip "depth cut"; round 4): the same entry points, other kernels
ENTRY_CUT = {
    "gs_render_tiles_backward_slab": ["gs::k_render_bwd<float, 1>", "gs::k_tile_order"],
    "gs_render_tiles_prefix": ["gs::k_render_fwd<float, 1>", "gs::k_bin_emit_buckets<2, 512>", "gs::k_tile_sort_overflow",
                               "gs::k_render_fwd_flagged<false>"],
    "gs_preprocess_forward": ["gs::k_preprocess<16, false>", "gs::k_cull_count", "gs::k_scan_counts<true>"],
    "gs_preprocess_backward": ["gs::k_preprocess_bwd<16>", "gs::k_preprocess_bwd<16, false>"],
    "gs_tile_count": ["gs::k_depth_hist", "gs::k_depth_colscan", "gs::k_depth_scatter", "gs::k_bin_count_buckets",
                      "gs::k_bin_colscan_cut<64>", "gs::k_scan_tiles_cut"],
    "gs_tile_emit_sort": ["gs::k_bin_emit_buckets<1, 1024>", "gs::k_tile_sort_runs"],
}
ENTRY = dict(ENTRY_CUT if "gs::k_bin_count_buckets" in vals else ENTRY_UNCUT)
if "gs::k_bwd_prologue" in vals:   # (round 4, late: slab clear + tile order as one launch behind its own entry point)
    ENTRY["gs_render_tiles_backward_slab"] = ["gs::k_render_bwd<float, 1>"]
    ENTRY["gs_render_backward_prologue"] = ["gs::k_bwd_prologue"]
res = {"workload": workload, "source": pmc_dir, "note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "entries": {}}
if "gs::k_bin_count_buckets" in vals:   # (round 4 on: also every kernel on its own)
    res["binning"] = "depth cut"
    res["kernels"] = {k: {"fetch_kib": v.get("FETCH_SIZE", 0.0), "write_kib": v.get("WRITE_SIZE", 0.0),
                          "hbm_bytes": int((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024)}
                      for k, v in sorted(vals.items()) if k.startswith("gs::")}
for entry, kernels in ENTRY.items():
    f = sum(vals.get(k, {}).get("FETCH_SIZE", 0.0) for k in kernels)
    w = sum(vals.get(k, {}).get("WRITE_SIZE", 0.0) for k in kernels)
    res["entries"][entry] = {"fetch_kib": f, "write_kib": w, "hbm_bytes": int((2 * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["entries"], indent=1))

if len(sys.argv) > 4:
    insts = {}
    for line in open(f"{pmc_dir}/pass1.txt"):
        m = re.match(r"(.{60}) (\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m and m.group(2) == "SQ_INSTS_VALU":
            insts[re.sub(r"\(.*", "", m.group(1).strip()).replace("void ", "")] = float(m.group(3))
    alias = {"gs_render_tiles_backward_slab": "gs_render_tiles_backward", "gs_render_tiles_prefix": "gs_render_tiles"}
    valu = {alias.get(e, e): int(sum(insts.get(k, 0.0) for k in ks)) for e, ks in ENTRY.items()}
    valu["_source"] = f"{pmc_dir}/pass1.txt (SQ_INSTS_VALU, wave-instructions per launch, workload {workload})"
    json.dump(valu, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(valu, indent=1))
