python tools/ab_pad.py --pads 0 256 2048 2304 --vlen 1e8 2>&1 | tail -5
python tools/ab_pad.py --pads 0 256 2304 --vlen 134217728 2>&1 | tail -4
python tools/ab_pad.py --pads 0 256 2304 --vlen 67108864 2>&1 | tail -4
python tools/ab_pad.py --pads 0 256 2048 2304 --vlen 1.25e7 2>&1 | tail -5
