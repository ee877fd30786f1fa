# Generates kangaroo/config.h from the reference's own template
# (include/kangaroo/config.h.in) with CMake's configure_file -- the same
# mechanism the reference build uses (CMakeLists.txt), with every optional
# library switched off (no Eigen/Assimp/Thrust/NPP/OpenCV in this image).
# Usage: cmake -DREF=/root/reference -DOUT=<path>/kangaroo/config.h -P gen_ref_config.cmake
set(_UNIX_ ON)
set(_LINUX_ ON)
set(_GCC_ ON)
set(CUDA_VERSION_MAJOR 12)
set(CUDA_VERSION_MINOR 8)
configure_file(${REF}/include/kangaroo/config.h.in ${OUT})
