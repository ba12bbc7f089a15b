# Tile-flow Cholesky beside the chain of launches (scripts/flow_check.py).  ONE size per process: the chain of launches runs
# first, in a process that does not hold the flow's streams yet (measured after a flow in the same process it loses 10 - 25 %:
# profiles/r06_flow.txt, section 9).  Then the flow alone under the lab build's plan switches (SR_FLOW_PANEL: block rows per
# panel, 100 = none; SR_FLOW_BAND: blocks next to the diagonal in 64-tiles).
for N in ${SIZES:-1000 1500 2000 2500 3000 3500 4000 5000 6000 8000 10000 14000}; do
  timeout 200 python scripts/flow_check.py $N 2>&1 | grep "^N=" | cut -c1-110
done
for cfg in "100 2" "6 2" "3 2" "8 2"; do set -- $cfg; SR_FLOW_ONLY=1 SR_FLOW_PANEL=$1 SR_FLOW_BAND=$2 timeout 300 python scripts/flow_check.py 2000 3000 5000 8000 2>&1 | grep "^N=" | cut -c1-90; done
