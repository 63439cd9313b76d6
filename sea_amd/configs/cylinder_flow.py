"""Configuration of the cylinder_flow case (reference configs/cylinder_flow.py): same function names, same keys and values for the data, mesh,
spatial-model and temporal-model entries; built from the shared table in _common.py."""
from ._common import spatial_config, temporal_config


def get_config_spatial():
    return spatial_config("cylinder_flow")


def get_config_temporal():
    return temporal_config("cylinder_flow")
