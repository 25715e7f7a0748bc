# dev tool: fwd+bwd us/step with the backward's halves serial (overlap=0) / forked (overlap=1), per workload
export HOST_PROFILE=0
for w in ${WORKLOADS:-c4_gdino_dec c2_q1k c2_q10k:Q=2000 c2_q10k:Q=3000 c2_q5k c4_gdino_dec:B=4 c4_gdino_dec:B=16 c4_gdino_dec:B=32 c4_gdino_dec:B=64}; do
  for o in 0 1; do
    echo "== $w overlap=$o: $(timeout -k 10 120 python tools/host_overhead.py $w 1000 overlap=$o 2>&1 | grep "us/step" | sed 's/.*: //' | tr '\n' '|')"
  done
done
