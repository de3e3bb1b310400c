echo "--- singles 40"; SINGLES=40 CALLS=20 python tools/batch_tail.py 2>/dev/null | tail -2
echo "--- profile + singles 40"; PROFILE=1 SINGLES=40 CALLS=20 python tools/batch_tail.py 2>/dev/null | tail -2
echo "--- torch + profile + singles 40"; TORCH=1 PROFILE=1 SINGLES=40 CALLS=20 python tools/batch_tail.py 2>/dev/null | tail -2
