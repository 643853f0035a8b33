#!/usr/bin/env bash
# tools/r06_forms.sh -- soak of the OTHER instantiations of the shading kernel (the closing session soaks <float, 2, 6>, the bench's): fp16 planes, a cone angle > 0 (MODE 1),
# the generic form (MODE 0), the fog scene; shipped library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06f; mkdir -p $O
soak() { l=$1; n=$2; shift 2; echo "== $l ($n renders) $*"; S=$SECONDS; env "$@" timeout 1700 python tools/repro_check.py $n 2>&1 | grep -v amdgpu.ids | tail -12; echo "wall $((SECONDS-S)) s"; }
{
soak fp16_planes 20000 REPRO_PLANES=float16
soak cone_angle_mode1 20000 REPRO_DT_GAMMA=0.0038095
soak generic_form_mode0 10000 SSDNERF_SHADE_GENERIC=1
soak fp16_planes_cone_angle 10000 REPRO_PLANES=float16 REPRO_DT_GAMMA=0.0038095
soak fog_scene_48_views 3000 REPRO_VARIANT=uniform REPRO_VIEWS=48
} > $O/soak_forms.txt 2>&1
cut -c1-300 $O/soak_forms.txt
