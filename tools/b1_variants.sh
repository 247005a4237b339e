#!/bin/bash
# Developer measurement (GPU box): the single-utterance path under chunk / lane settings.
for v in "" "PK_MI355_CHUNK=512 PK_MI355_LANES=2" "PK_MI355_CHUNK=256 PK_MI355_LANES=2" "PK_MI355_CHUNK=512 PK_MI355_LANES=1"; do
  echo "== $v"
  env $v python3 tools/b1_probe.py 2>&1 | grep -v "^/opt"
done
