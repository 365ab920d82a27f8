cp keynet_amd/libkeynet_hip.so /tmp/new.so; cp keynet_amd/libkeynet_hip_old.so /tmp/old.so
run() { for cfg in "64 64 224" "128 128 112" "256 256 56" "512 512 28" "512 512 14"; do set -- $cfg; timeout 200 python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --perm --iters 9 2>&1 | tail -1 | cut -c1-125; done; }
for rep in 1 2; do
echo "== new"; cp /tmp/new.so keynet_amd/libkeynet_hip.so; run
echo "== old"; cp /tmp/old.so keynet_amd/libkeynet_hip.so; run
done
cp /tmp/new.so keynet_amd/libkeynet_hip.so
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
