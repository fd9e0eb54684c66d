#!/bin/bash
{
for mb in 0 32 256; do echo "== mailbox MB $mb"; NYX_HIP_MAILBOX_MB=$mb timeout 200 python tools/sweep.py 2 10000 3 '{"new":{}}' 2 | grep "^new"; done
} 2>&1 | grep -v amdgpu
