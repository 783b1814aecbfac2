#!/bin/bash
# Run in the BUILD container: the round's committed profile set - per configuration one gpurun call per profiler session
# (plain run + kernel trace, then the three counter passes), condensed into profiles/<tag>_* and profiles/traffic.json.
#   usage: tools/collect_round.sh r03 [config ...]     configs: wn_b1 wn_b4 wn_b8 ac_b1 wn_ragged lynx_b8 var_b8 wn_b8_x3 lynx_b8_x3
set -o pipefail
rt=$1; shift
cfgs=${@:-"wn_b1 wn_b8 wn_ragged lynx_b8 var_b8"}
for c in $cfgs; do
  case $c in
    wn_b1)     tag=${rt}_wavenet_dpm50_b1;        key=wavenet_dpm50/B1/T1000;         args="--workload wavenet_dpm50 --batch 1 --steps 20 --warmup 3" ;;
    wn_b4)     tag=${rt}_wavenet_dpm50_b4;        key=wavenet_dpm50/B4/T1000;         args="--workload wavenet_dpm50 --batch 4 --steps 8 --warmup 2" ;;
    wn_b8)     tag=${rt}_wavenet_dpm50_b8;        key=wavenet_dpm50/B8/T1000;         args="--workload wavenet_dpm50 --batch 8 --steps 6 --warmup 2" ;;
    wn_ragged) tag=${rt}_wavenet_dpm50_ragged_b8; key=wavenet_dpm50_ragged/B8/T1536;  args="--workload wavenet_dpm50 --batch 8 --ragged --steps 6 --warmup 2" ;;
    lynx_b8)   tag=${rt}_lynxnet_ddim100_b8;      key=lynxnet_ddim100/B8/T1000;       args="--workload lynxnet_ddim100 --batch 8 --steps 3 --warmup 1" ;;
    wn_b8_x3)  tag=${rt}_wavenet_dpm50_bf16x3_b8; key=wavenet_dpm50_bf16x3/B8/T1000;  args="--workload wavenet_dpm50 --batch 8 --precision bf16x3 --steps 6 --warmup 2" ;;
    lynx_b8_x3) tag=${rt}_lynxnet_ddim100_bf16x3_b8; key=lynxnet_ddim100_bf16x3/B8/T1000; args="--workload lynxnet_ddim100 --batch 8 --precision bf16x3 --steps 3 --warmup 1" ;;
    ac_b1)     tag=${rt}_acoustic_default_b1;     key=acoustic_default/B1/T1000;      args="--workload acoustic_default --batch 1 --steps 10 --warmup 2" ;;
    var_b8)    tag=${rt}_variance_reflow20_b8;    key=variance_reflow20/B8/T1000;     args="--workload variance_reflow20 --batch 8 --steps 6 --warmup 2" ;;
    *) echo "unknown config $c"; exit 2 ;;
  esac
  bash tools/collect_all.sh $tag "both FETCH_SIZE WRITE_SIZE MFMA" $args || exit $?
  d=gpurun_out/prof_$tag
  python tools/summarize_profile.py $tag $d/trace $d/FETCH_SIZE $d/WRITE_SIZE $d/MFMA --traffic $key - --bench $d || exit $?
done
