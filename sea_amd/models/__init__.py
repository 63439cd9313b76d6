"""Host-side mirrors of the reference model classes; every forward runs a pre-built launch plan of the HIP library."""
