"""HBM-side traffic per launch from three rocprofv3 --pmc result directories (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum+TCC_MISS_sum,
each collected in its own run of tools/run_kernels_once.py): prints one JSON object, gfx950 corrections applied
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half of the bytes of wide coalesced reads -> doubled)."""
import glob, json, sqlite3, sys, collections


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


fetch, write, tcc = (per_kernel(d) for d in sys.argv[1:4])
names = {"fgvc_pair_topk_bf16x4": "pair_topk_kernel_v4", "fgvc_pair_topk_f32": "pair_topk_kernel_v3", "fgvc_corr_volume_bf16x3": "corr_volume_bf16_kernel<256, 3",
         "fgvc_corr_volume_bf16": "corr_volume_bf16_kernel<256, 1", "fgvc_corr_volume_f32": "corr_volume_f32_kernel", "fgvc_conv_split_f32": "conv_split_kernel<3, 256",
         "fgvc_stem7_split_f32": "stem7_kernel", "fgvc_conv_s2_split_f32": "conv_s2_kernel<3>"}
res = {}
for key, sub in names.items():
    def avg(tbl, cn):
        vals = [v for k, d in tbl.items() if sub in k for v in d.get(cn, [])]
        return sum(vals) / len(vals) if vals else None
    fs, ws, hit, miss = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE"), avg(tcc, "TCC_HIT_sum"), avg(tcc, "TCC_MISS_sum")
    if fs is None or ws is None:
        continue
    res[key] = {"kernel": sub, "FETCH_SIZE_KB_raw": fs, "WRITE_SIZE_KB_raw": ws, "fetch_bytes_corrected": 2 * fs * 1024, "write_bytes": ws * 1024,
                "hbm_bytes_per_launch": 2 * fs * 1024 + ws * 1024, "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss else None}
print(json.dumps(res, indent=1))
