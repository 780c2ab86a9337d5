"""Output keys of the fields (same string values as neusky/field_components/neusky_fieldheadnames.py:6-14,
plus the nerfstudio FieldHeadNames members the path touches: SDF, NORMALS, GRADIENT, ALPHA)."""
from enum import Enum


class NeuSkyFieldHeadNames(Enum):
    ALBEDO = "albedo"
    VISIBILITY = "visibility"
    ILLUMINATION_DIRECTION = "illumination_direction"
    TERMINATION_DISTANCE = "termination_distance"
    SHININESS = "shininess"
    PROBABILITY_OF_HIT = "probability_of_hit"


class FieldHeadNames(Enum):
    """subset of nerfstudio.field_components.field_heads.FieldHeadNames used on the hot path"""
    SDF = "sdf"
    NORMALS = "normals"
    GRADIENT = "gradient"
    ALPHA = "alpha"
    DENSITY = "density"
