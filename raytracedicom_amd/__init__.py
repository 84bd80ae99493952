"""raytracedicom_amd — MI355X-native pencil-beam proton dose engine (drop-in for the reference's
cudaWrapperProtons hot path, src/kernel_wrapper.cu:381-1369). See DESIGN.md and include/rtd.h."""
__version__ = "0.1.0"
