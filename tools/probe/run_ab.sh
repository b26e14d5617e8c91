mkdir -p gpurun_out/r2m
for s in 12 13 14 15 16 17; do ORL_LIB_VARIANT=alt timeout 1200 python3 tools/fuzz_cross.py $s 30 > gpurun_out/r2m/fuzz$s.txt 2>&1; tail -1 gpurun_out/r2m/fuzz$s.txt; grep -c "nsfnet_chen.*320" gpurun_out/r2m/fuzz$s.txt; grep -v OK gpurun_out/r2m/fuzz$s.txt | head -5; done
