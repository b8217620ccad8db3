for b in 160 256 320 512; do echo "== wgrad blocks $b"; S2AG_BF16_WGRAD_BLOCKS=$b MODE=bf16 python tools/run_cfg4.py 2>&1 | tail -1 | cut -c100-200; done
