#!/usr/bin/env bash
# The reference's own programs, compiled UNCHANGED against the facade (oracle/ref_tests, oracle/ref_examples), on the GPU box: exit code, wall time and
# closing lines of every test binary and example, cold (fresh model-cache folder) and -- for the examples that compile functions at run time -- warm.
# Output: gpurun_out/reference_programs.log  (tools/collect_round.sh copies it to profiles/<tag>_reference_programs.log).
set -uo pipefail
mkdir -p gpurun_out
out=gpurun_out/reference_programs.log
: > $out
run() {  # name, folder, pass
  local exe=oracle/_ref/$1 dir=$2
  [ -x $exe ] || { echo "$1: not built" | tee -a $out; return; }
  mkdir -p $dir
  local t0=$(date +%s%N)
  (cd $dir && UNGAR_CODEGEN_FOLDER=$dir timeout 1500 $OLDPWD/$exe > $dir/$1.out 2>&1); local rc=$?
  local t1=$(date +%s%N)
  local ms=$(( (t1 - t0) / 1000000 ))
  printf "%-28s %s  rc %d  wall %d.%03d s  | %s\n" "$1" "$3" $rc $((ms / 1000)) $((ms % 1000)) "$(grep -v "^\s*$" $dir/$1.out | tail -1 | cut -c1-110)" | tee -a $out
}
for t in ref_variable_test ref_variable_test_eigen ref_utils_test_eigen ref_function_test_eigen ref_soft_sqp_test_eigen ref_robot_test_eigen; do run $t /tmp/refprog/$t cold; done
for e in variable_example variable_map_example function_example quadrotor_example rc_car_example quadruped_example; do
  run $e /tmp/refprog/$e cold
  run ${e}_eigen /tmp/refprog/${e}_eigen cold
done
run quantity_example_eigen /tmp/refprog/quantity cold
run robot_example_eigen /tmp/refprog/robot cold
for e in quadrotor_example rc_car_example quadruped_example robot_example_eigen; do run $e /tmp/refprog/$([ $e = robot_example_eigen ] && echo robot || echo $e) "warm"; done
