for d in 0 1 2 4 8 16 5 7 15 31; do echo "dbg=$d"; HCONV_DBG=$d timeout -k 5 60 python3 scripts/bench_conv.py --shapes L1b,L2b,L2c --only fwd 2>&1 | grep -E "^L"; done
