timeout 600 tools/hbm_probe 1e8 0 tr 2>&1 | tail -80
