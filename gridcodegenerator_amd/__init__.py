"""MI355X-native rigid-body-dynamics code generator (gfx950 / CDNA4, HIP, wave64).

``from gridcodegenerator_amd import GRiDCodeGenerator`` mirrors the reference package import
(reference ``__init__.py:1``).
"""
from .GRiDCodeGenerator import GRiDCodeGenerator  # noqa: F401
from .urdf import URDFParser, load_urdf, robot_to_urdf  # noqa: F401  (URDF -> robot object, the step before the generator)
