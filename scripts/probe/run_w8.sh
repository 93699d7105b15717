cd scripts/probe
export GS_TRACE_QUIET=1
for L in "0 8 16 128 256 256" "0 8 32 256 128 128" "0 8 64 512 64 64" "0 8 8 64 256 256"; do
  echo "== $L"
  for b in def m0_w8a m0_w8b m0_w8c m0_w8d m0_w8e; do printf "%-10s " $b; timeout 60 ./igemm_trace_$b $L 10 2>&1 | head -1; done
done
for L in "2 8 16 128 256 128" "2 8 32 256 128 64" "2 8 8 64 256 256"; do
  echo "== $L"
  for b in def m2_w8a m2_w8b; do printf "%-10s " $b; timeout 60 ./igemm_trace_$b $L 10 2>&1 | head -1; done
done
