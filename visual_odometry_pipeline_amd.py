"""Import alias: the package directory is named `visual-odometry-pipeline_amd`, which is not a
valid Python identifier; `import visual_odometry_pipeline_amd` resolves to it."""
import importlib
import sys

_pkg = importlib.import_module("visual-odometry-pipeline_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("visual-odometry-pipeline_amd."):
        sys.modules[__name__ + _name[len("visual-odometry-pipeline_amd"):]] = _mod
sys.modules[__name__] = _pkg
