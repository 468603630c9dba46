# which feature makes the bitwise guard fire?  verify every chunk on the first ViT blocks,
# one run from the model's first matrix as the harness does it
R=$GRAFT_REPO_ROOT
L=$(python3 -c "print(','.join(str(i) for i in range(0,20)))")
for cfg in "1 2" "0 2" "1 0" "0 0"; do
  set -- $cfg
  echo "== BATCHED_ADVANCE=$1 PAD_SLOTS=$2"
  ECOFLAP_VERIFY_BATCHED=1 ECOFLAP_DEBUG_BATCHED=1 ECOFLAP_BATCHED_ADVANCE=$1 ECOFLAP_PAD_SLOTS=$2 python3 $R/bench.py --no-cpu-baseline --warmup 0 --layers $L 2>/dev/null | grep -v "^{" | head -5
  ECOFLAP_VERIFY_BATCHED=1 ECOFLAP_BATCHED_ADVANCE=$1 ECOFLAP_PAD_SLOTS=$2 python3 $R/bench.py --no-cpu-baseline --warmup 0 --layers $L 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); sf=d['breakdown']['suffix_forward']
print({k:v for k,v in sf.items() if 'disabled' in k or 'mismatch' in k or 'grouped' in k or 'checks' in k or 'padded' in k})"
done
