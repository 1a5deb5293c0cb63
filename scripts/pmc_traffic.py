"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per
pass) over `bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0`.

    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db > profiles/rNN_pmc_traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming
reads at 64 bytes and is doubled (checked here on adam_kernel, whose traffic is known exactly:
4 reads and 3 writes of the flat parameter buffer); WRITE_SIZE is exact."""
import json
import sqlite3
import sys

# substring of the profiler's kernel name -> the name bench.py reports for it
NAMES = {"dconv_kernel<4, false>": "dconv_256x64", "dconv_kernel<4, true>": "dconv_256x64_rp",
         "dconv_kernel<2, false>": "dconv_256x32", "dconv_kernel<2, true>": "dconv_256x32_rp",
         "dconv_kernel<1, false>": "dconv_256x16", "dconv_kernel<1, true>": "dconv_256x16_rp",
         "hwgrad_kernel<64,": "hwgrad_64x32x9", "hwgrad_kernel<32,": "hwgrad_32x32x9", "adam_kernel": "adam",
         "pconv_kernel<__bf16, 4>": "pconv_64x64_k4", "pconv_kernelIDF16bLi4": "pconv_64x64_k4",
         "maxpool2_bwd": "maxpool2_bwd", "maxpool2_fwd": "maxpool2_fwd", "tconv_fwd": "tconv_fwd", "tconv_dgrad": "tconv_dgrad",
         "finalize": "finalize", "head_bwd": "head_bwd", "head_fwd": "head_fwd", "nchw_to_nhwc": "nchw_to_nhwc"}


def per_kernel(db, counter):
    """short name -> (mean bytes per launch, launches), the variants that share a short name pooled."""
    c = sqlite3.connect(db)
    acc = {}
    for k, v, n in c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
                             "group by kernel_name", (counter,)):
        for pat, short in NAMES.items():
            if pat in k:
                tot, cnt = acc.get(short, (0.0, 0))
                acc[short] = (tot + v * 1024.0 * n, cnt + n)
                break
    return {k: (tot / cnt, cnt) for k, (tot, cnt) in acc.items()}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {"unit": "bytes per launch (mean over the launches of the kernel in the run)",
           "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE x 1", "kernels": {}}
    for k in sorted(f):
        if k in w:
            res["kernels"][k] = {"read": round(2 * f[k][0]), "written": round(w[k][0]), "traffic": round(2 * f[k][0] + w[k][0]),
                                 "launches_sampled": f[k][1]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
