#!/bin/bash
# Rebuild one of the measured-and-dropped kernel variants as it was measured: a worktree of the commit the variant was
# derived from, the variant's patch applied to that commit's kernel, built there.  Run in the development container (it needs
# the git history; the GPU box has none), then point gpurun at the worktree's library:
#   tools/ubench/variants/make_variant.sh viterbi_split_barrier_kernel        -> /tmp/nc_variant_<name>/nanocall_amd/libnanocall_hip.so
# The shipped tree is never patched (an interrupted in-place patch left a half-patched kernel behind).
set -euo pipefail
name=$1
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../../.." && pwd)
patch_file=$here/$name.patch
[ -f "$patch_file" ] || { echo "no such variant: $name" >&2; exit 2; }
# "--- a/nanocall_amd/csrc/<file> (<commit>)"
read -r file commit < <(sed -n '1s|^--- a/\([^ ]*\) (\([0-9a-f]*\)).*|\1 \2|p' "$patch_file")
wt=/tmp/nc_variant_$name
rm -rf "$wt"; git -C "$root" worktree prune
git -C "$root" worktree add -q --detach "$wt" "$commit"
patch -s "$wt/$file" "$patch_file"
make -s -C "$wt/nanocall_amd/csrc" -j8
echo "$wt/nanocall_amd/libnanocall_hip.so  (commit $commit + $name.patch)"
