#!/bin/bash
# Pin the Sinkhorn oracle against the REAL geomloss 0.2.4 (the one piece of this repo whose parity is unpinned: the package is
# absent from the reference tree, from this image's offline wheelhouse and from every GPU box -- DESIGN.md section 1).
# Needs a network route to PyPI and the reference checkout at /root/reference.  One command:
#
#     bash tools/pin_geomloss.sh
#
# What it does: a throw-away venv that sees the system torch / numpy, `pip install geomloss==0.2.4` (pure Python; pykeops is
# optional and not used by the tensorized backend the reference calls), regenerates tests/golden/*.npz with the real package
# behind the reference's own AllPairMaskedWasserstein wrapper (tests/golden/make_golden.py picks the real package up when it
# imports; ASPIRE_REQUIRE_GEOMLOSS makes it refuse the stand-in), and runs the CPU oracle tests against the new fixtures --
# tests/test_oracle_cpu.py compares the solver-dependent outputs at 1e-4 once scores.npz says solver = geomloss-0.2.4.
# If they pass, commit tests/golden/scores.npz and change "parity unpinned" to "pinned" in oracle/aspire_oracle.py's header and
# DESIGN.md section 1.  If they fail, the restatement in oracle/aspire_oracle.py (geomloss_sinkhorn_tensorized) is what to fix.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VENV=${VENV:-/tmp/aspire_geomloss_venv}
test -d /root/reference || { echo "need the reference checkout at /root/reference"; exit 2; }
python3 -m venv --system-site-packages "$VENV"
PIP_NO_INDEX= PIP_INDEX_URL=${PIP_INDEX_URL:-https://pypi.org/simple} "$VENV/bin/pip" install --no-deps "geomloss==0.2.4"
"$VENV/bin/python" -c "import geomloss; print('geomloss', geomloss.__version__)"
cd "$ROOT"
ASPIRE_REQUIRE_GEOMLOSS=1 "$VENV/bin/python" tests/golden/make_golden.py
"$VENV/bin/python" -m pytest tests/test_oracle_cpu.py -x -q -s
echo "solver pinned: commit tests/golden/*.npz and update the 'parity unpinned' notes (oracle/aspire_oracle.py, DESIGN.md section 1)"
