#!/bin/bash
# Same-box A/B of two states of cudabrot_amd/csrc (a GPU box has no .git, and boxes differ by +-4 %).
#   tools/ab_export.sh <name> [<git-ref>]   copy csrc of <git-ref> (default: the working tree) to tools/_ab/<name>/
# then, in ONE gpurun call:  ./tools/gpu_ab_run.sh <name1> <name2> <name1> <name2>
# tools/_ab/ is scratch: delete it afterwards (it is git-ignored).
set -eu
name=$1; ref=${2:-}
dst=tools/_ab/$name
rm -rf "$dst"; mkdir -p "$dst"
for f in $(git ls-files cudabrot_amd/csrc | grep -v Makefile); do
  if [ -n "$ref" ]; then git show "$ref:$f" > "$dst/$(basename "$f")"; else cp "$f" "$dst/"; fi
done
echo "exported $(ls "$dst" | wc -l) files of ${ref:-the working tree} to $dst"
