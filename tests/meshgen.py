"""The procedural scene generator lives in the package (renderer_amd/meshgen.py: bench.py's mesh1m workload
uses it too); the tests keep importing it under its old name, private helpers included."""
import sys

import renderer_amd.meshgen as _impl

sys.modules[__name__] = _impl
