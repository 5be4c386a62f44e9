#!/bin/bash
# Per-phase cycle breakdown of the general Forward pipeline (needs a trace build:
#   make -C historian_amd/csrc EXTRA=-DHX_DAG_TRACE OBJDIR=../build_abl1 LIBDIR=../lib_abl1)
export HX_LIB_PATH=$GRAFT_REPO_ROOT/historian_amd/lib_abl1/libhistorian_hip.so
timeout -k 10 200 python tools/dag_bench.py 1 fwdonly 2>&1 | grep -E "^trace|fast|exact|band" 
