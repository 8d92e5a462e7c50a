for seed in 61 62 63 64; do timeout 900 python tools/fuzz_parity.py 300 $seed 2>&1 | grep -v '^\.\.\|^ok' | cut -c1-900 | tail -3; done
for seed in 65 66; do timeout 900 python tools/fuzz_parity.py 300 $seed 0 2 2>&1 | grep -v '^\.\.\|^ok' | cut -c1-900 | tail -3; done
