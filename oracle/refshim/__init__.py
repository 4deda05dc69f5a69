"""Stand-ins (mini-xarray, absl stubs) that let the REFERENCE's own Python
modules import and run in this container; see xarray/__init__.py.  Test
infrastructure only -- never imported by weatherbench2_amd/."""
