# Tile-flow Cholesky beside the chain of launches on ONE box (scripts/flow_check.py), then the flow alone under the lab build's
# plan switches (SR_FLOW_PANEL: block rows per panel, 100 = none; SR_FLOW_BAND: blocks next to the diagonal in 64-tiles).
S="1500 2000 2500 3000 3500 4000 5000"
timeout 200 python scripts/flow_check.py $S 2>&1 | grep "^N=" | cut -c1-70
for cfg in "100 2" "6 2" "3 2" "100 1"; do set -- $cfg; SR_FLOW_ONLY=1 SR_FLOW_PANEL=$1 SR_FLOW_BAND=$2 timeout 200 python scripts/flow_check.py $S 2>&1 | grep "^N=" | cut -c1-90; done
