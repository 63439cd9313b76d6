"""Configuration of the multiphase_flow case (reference configs/multiphase_flow.py): same function names, same keys and values for the data, mesh,
spatial-model and temporal-model entries; built from the shared table in _common.py."""
from ._common import spatial_config, temporal_config


def get_config_spatial():
    return spatial_config("multiphase_flow")


def get_config_temporal():
    return temporal_config("multiphase_flow")
