#!/bin/bash
# GPU box: the post-process under two settings of ONE environment switch, alternating: parity tests, device time of the call on the stress
# maps, kernel stats.  usage: post_env_ab.sh VAR "a b ..."
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in $2; do
  export $1=$v
  echo "== $1=$v"
  python tools/dbg/post_graph_ab.py 2>&1 | tail -1
  bash tools/prof_post.sh r6_env_$v | grep -v "calls    [12] avg"
done
