"""Turns the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_passes.sh into profiles/<name>.json:
per-kernel HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> doubled; both are in KiB).
usage: python scripts/make_traffic_json.py <pmc_dir> <out.json> <workload>"""
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
ENTRY = {
    "gs_render_tiles_backward_slab": ["gs::k_render_bwd<float, 1>"],
    "gs_render_tiles_prefix": ["gs::k_render_fwd<float, 1>", "gs::k_render_fwd_flagged", "gs::k_tile_sort_flagged<8192>",
                               "gs::k_tile_sort_flagged<4096>"],
    "gs_preprocess_forward": ["gs::k_preprocess<16, false>", "gs::k_cull_count", "gs::k_scan_counts", "gs::k_camera_center"],
    "gs_preprocess_backward": ["gs::k_preprocess_bwd<16>"],
    "gs_tile_count": ["gs::k_bin_count", "gs::k_bin_colscan", "gs::k_scan_tiles"],
    "gs_tile_emit_sort": ["gs::k_bin_emit", "gs::k_tile_sort<true>", "gs::k_tile_sort_big<true>"],
}
res = {"workload": workload, "source": pmc_dir, "note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "entries": {}}
for entry, kernels in ENTRY.items():
    f = sum(vals.get(k, {}).get("FETCH_SIZE", 0.0) for k in kernels)
    w = sum(vals.get(k, {}).get("WRITE_SIZE", 0.0) for k in kernels)
    res["entries"][entry] = {"fetch_kib": f, "write_kib": w, "hbm_bytes": int((2 * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["entries"], indent=1))
