#!/bin/bash
O=gpurun_out/r4_diag2; mkdir -p $O
{
timeout 240 python tools/sweep.py 2 10000 3 '{"base":{}, "base_prof":{"profile":1}, "noharm_prof":{"debug_flags":512,"profile":1}, "feed1":{"harmonics_feed":1}, "feed1_prof":{"harmonics_feed":1,"profile":1}, "frac25":{"coop_fraction":0.25}, "frac30":{"coop_fraction":0.30}, "frac33":{"coop_fraction":0.33}, "cols12":{"coop_max_columns":12}, "cols10":{"coop_max_columns":10}, "cal":{"schedule":1}, "cal_feed1":{"schedule":1,"harmonics_feed":1}}'
timeout 240 python tools/sweep.py 5 6250 1 '{"base":{}, "base_prof":{"profile":1}, "r155":{"coop_helper_ratio":1.55}, "r155_c56":{"coop_helper_ratio":1.55,"coop_max_columns":56}, "r155_c70":{"coop_helper_ratio":1.55,"coop_max_columns":70}, "r155_c84":{"coop_helper_ratio":1.55,"coop_max_columns":84}, "r155_f45":{"coop_helper_ratio":1.55,"coop_fraction":0.45}, "r13":{"coop_helper_ratio":1.3}, "r155_prof":{"coop_helper_ratio":1.55,"profile":1}}'
} > $O/log.txt 2>&1
cat $O/log.txt
