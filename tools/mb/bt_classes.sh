for cls in HLHLHL HHNNLL HHHHLL HLHLLL; do
for torch in "" 1; do
echo "--- classes $cls torch=$torch"; RV_BATCH_CLASSES=$cls TORCH=$torch CALLS=24 python tools/batch_tail.py 2>/dev/null | tail -1
done; done
