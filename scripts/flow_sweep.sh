# sweep of the tile flow's lab switches (scripts/flow_check.py; lab build): panel width, band
for cfg in ${CFGS:-"100 2" "100 1" "4 2" "6 2" "4 1"}; do
  set -- $cfg
  SR_FLOW_ONLY=1 SR_FLOW_PANEL=$1 SR_FLOW_BAND=$2 timeout 100 python scripts/flow_check.py ${SIZES:-2000 3000 5000} 2>&1 | grep "^N="
done
