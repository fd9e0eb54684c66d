#!/bin/bash
for c in 2 3 4; do CFG=$c bash tools/profile_round.sh 2>&1 | tail -22; done
CFG=5 STEPS=1 bash tools/profile_round.sh 2>&1 | tail -22
