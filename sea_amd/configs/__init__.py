"""Case configurations named after the reference's configs (cylinder_flow, multiphase_flow): `get_config(flow_type, model_type)` resolves
them the way the reference CLI does (main.py:23-29)."""
import importlib


def get_config(flow_type: str, model_type: str = "temporal"):
    mod = importlib.import_module(f"{__name__}.{flow_type}")
    return mod.get_config_temporal() if model_type == "temporal" else mod.get_config_spatial()
