"""rocprofv3 --pmc result directories (one counter group per run of tools/run_kernels_once.py; PMC passes must not be combined with
traces) -> one JSON object per kernel: HBM-side traffic (gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE counts half of the
bytes of wide coalesced reads -> doubled; both are in KB), L2 hit rate, and matrix-pipe utilisation

    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)

(SQ_VALU_MFMA_BUSY_CYCLES counts pipe-busy cycles summed over the chip's 1024 SIMDs: 32 per v_mfma_f32_32x32x16_bf16;
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so / 8 = the launch's duration in shader clocks).  The counter is cross-checked below
against the MFMA count the dense volume kernels are known to execute.

    python tools/pmc_report.py DIR_FETCH DIR_WRITE DIR_TCC DIR_MFMA > profiles/r02_pmc.json
"""
import collections, glob, json, sqlite3, sys


def per_kernel(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*results.db", recursive=True):
        c = sqlite3.connect(f).cursor()
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        tab = [t for t in tabs if t.startswith("counters_collection")][0]
        rows = c.execute(f"select dispatch_id, kernel_name, counter_name, sum(value) from {tab} group by dispatch_id, counter_name").fetchall()
        for disp, k, cn, v in rows:
            out[k][cn].append(v)
    return out


fetch, write, tcc, mfma = (per_kernel(d) for d in sys.argv[1:5])
names = {"fgvc_pair_topk_f16f6": "pair_topk_kernel_v7", "fgvc_pair_topk_f16f6[v8, opt-in]": "pair_topk_kernel_v8",
         "fgvc_local_corr_topk_f16x3[merge]": "local_merge_kernel", "fgvc_c2f_refine_f32": "c2f_refine_kernel", "fgvc_pair_topk_f16x3": "pair_topk_kernel_v6", "fgvc_pair_topk_f32": "pair_topk_kernel_v2",
         "fgvc_corr_volume_f16f6": "corr_volume_f16f6_kernel", "fgvc_corr_volume_f16f8": "corr_volume_f16f8_v2_kernel", "fgvc_corr_volume_bf16x3": "corr_volume_bf16_kernel<256, 3",
         "fgvc_corr_volume_bf16": "corr_volume_bf16_kernel<256, 1", "fgvc_corr_volume_f32": "corr_volume_f32_kernel",
         "fgvc_conv_split_f32": "conv_split_kernel<3, 256, 1, 3, 4, true, 0, 2, false, false", "fgvc_conv_split_fmt_f32[f16f8]": "conv_split_kernel<3, 256, 1, 3, 4, false, 1, 2, false, false",
         "fgvc_conv_split_fmt_f32[f16f6]": "conv256p_kernel<256, 3, false, false, false",          # (round 5: the one-wave-per-SIMD stream kernel)
         "fgvc_conv_split_proj_fmt_f32[f16f6]": "conv256p_kernel<256, 3, false, true, false",
         "fgvc_conv_split_bank_f16f6p_f32[f16f6]": "conv256p_kernel<256, -1, false, false, true, true",
         "fgvc_merge_refine_topk_f32[mark]": "merge_mark_kernel", "fgvc_merge_refine_topk_f32[refine]": "refine_kernel",
         "fgvc_merge_refine_topk_f32[scan]": "refine_scan_kernel",
         "fgvc_conv64_split_f32": "conv64_kernel", "fgvc_conv64_split_fmt_f32[conv1]": "conv64p_kernel<false, false, 1>",
         "fgvc_conv64_split_fmt_f32[conv2]": "conv64p_kernel<true, true, 1>",
         "fgvc_stem7_split_f32": "stem7_kernel", "fgvc_conv_s2_split_f32": "conv_s2_kernel<3>"}
HW = 120 * 214
tiles = -(-HW // 32) * -(-HW // 32)
# matrix-pipe cycles the dense kernels execute by construction (ragged tiles included), for the cross-check
expect = {"fgvc_corr_volume_bf16x3": tiles * 48 * 32, "fgvc_corr_volume_bf16": tiles * 16 * 32,
          "fgvc_corr_volume_f16f6": tiles * 768, "fgvc_corr_volume_f16f8": tiles * 1024, "fgvc_corr_volume_f32": tiles * 128 * 64}
res = {}
for key, sub in names.items():
    def avg(tbl, cn):
        vals = [v for k, d in tbl.items() if sub in k for v in d.get(cn, [])]
        return sum(vals) / len(vals) if vals else None
    fs, ws, hit, miss = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE"), avg(tcc, "TCC_HIT_sum"), avg(tcc, "TCC_MISS_sum")
    busy, gui, sqb = avg(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(mfma, "GRBM_GUI_ACTIVE"), avg(mfma, "SQ_BUSY_CYCLES")
    if fs is None and busy is None:
        continue
    r = {"kernel": sub}
    if fs is not None and ws is not None:
        r.update({"FETCH_SIZE_KB_raw": fs, "WRITE_SIZE_KB_raw": ws, "fetch_bytes_corrected": 2 * fs * 1024, "write_bytes": ws * 1024,
                  "hbm_bytes_per_launch": 2 * fs * 1024 + ws * 1024})
    if hit is not None and miss:
        r["l2_hit_rate"] = hit / (hit + miss)
    if busy is not None and gui:
        r.update({"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "SQ_BUSY_CYCLES": sqb,
                  "launch_shader_clocks": gui / 8, "mfma_util": busy / (gui / 8 * 1024)})
        if key in expect:
            r["mfma_cycles_by_construction"] = expect[key]
            r["counter_over_construction"] = busy / expect[key]
    res[key] = r
print(json.dumps(res, indent=1))
