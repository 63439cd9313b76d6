"""Model / dataset configurations named after the reference configs (cylinder_flow, multiphase_flow)."""
