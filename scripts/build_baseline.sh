#!/bin/bash
# Builds an EARLIER round's libyolo2hip.so from the git history into profiles/baseline/libyolo2hip_<tag>.so, for same-box A/B runs
# (scripts/gpu_evidence.sh section `rounds`; select it per process with YOLO2_LIB_PATH=... YOLO2_LIB_BASELINE=1).
#   usage: bash scripts/build_baseline.sh r05 1bd4f74
# A build artefact (git-ignored like every .so; it travels to the GPU box with the gpurun snapshot).  The commit it was built from is recorded
# next to it.  Run HERE, before the gpurun call (about a minute of host time).
set -e
TAG=${1:?tag, e.g. r05}; COMMIT=${2:?commit}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$COMMIT" yolo_tf_amd/csrc include | tar -x -C "$TMP"
( cd "$TMP/yolo_tf_amd/csrc" && python build.py > /dev/null )
mkdir -p "$ROOT/profiles/baseline"
cp "$TMP/yolo_tf_amd/csrc/libyolo2hip.so" "$ROOT/profiles/baseline/libyolo2hip_$TAG.so"
git -C "$ROOT" rev-parse "$COMMIT" > "$ROOT/profiles/baseline/libyolo2hip_$TAG.commit"
ls -la "$ROOT/profiles/baseline/"
