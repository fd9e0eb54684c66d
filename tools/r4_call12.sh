#!/bin/bash
{
timeout 300 python tools/sweep.py 5 6250 1 '{"auto2":{}, "one_part":{"debug_flags":524288}, "p2_f45":{"coop_fraction":0.45}, "p2_f50":{"coop_fraction":0.50}, "p2_f60":{"coop_fraction":0.60}, "p2_f50_r16":{"coop_fraction":0.50,"coop_helper_ratio":1.6}, "auto2_prof":{"profile":1}}' 1 64
timeout 300 python tools/sweep.py 2 10000 3 '{"new":{}, "two_parts":{"debug_flags":262144}, "two_parts_c20":{"debug_flags":262144,"coop_max_columns":20,"coop_fraction":0.5}, "two_parts_c24":{"debug_flags":262144,"coop_max_columns":24,"coop_fraction":0.5},"two_parts_c28":{"debug_flags":262144,"coop_max_columns":28,"coop_fraction":0.6}}' 1 64
} 2>&1 | grep -v amdgpu.ids
