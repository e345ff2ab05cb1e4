"""modle_amd -- MI355X-native loop-extrusion simulation core (drop-in for MoDLE's
Simulation::simulate_one_cell path).  See DESIGN.md and include/modle_hip.h."""
from .params import CellResult, Config, Task  # noqa: F401
