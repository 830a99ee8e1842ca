#!/bin/bash
# Run in the build container, from the repo root:   bash profiles/run_collect.sh <tag> [gpurun timeout seconds]
# Refuses a dirty tree (the profiles must describe a commit), stamps the revision into .build_head, rebuilds the
# library from that revision and runs profiles/collect.sh on a GPU box.  Afterwards: copy gpurun_out/<tag>/<tag>_* into
# profiles/ and point profiles/LATEST at them.
set -eu
tag=${1:?tag}
limit=${2:-1500}
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then
    echo "uncommitted changes: commit first, the profiles are stamped with HEAD" >&2
    exit 2
fi
git rev-parse HEAD > .build_head
make -C enzymm_amd/csrc -B all > /dev/null
make -C oracle all > /dev/null
/usr/local/graft/bin/gpurun --timeout "$limit" -- "bash profiles/collect.sh $tag"
