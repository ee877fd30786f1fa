// kangaroo.h -- umbrella include of the MI355X KinectFusion volumetric path (counterpart of the
// reference's include/kangaroo/kangaroo.h:18-44, restricted to the hot path of SURVEY.md section 8).
#pragma once

#include <kangaroo/config.h>
#include <kangaroo/platform.h>

#include <kangaroo/Memory.h>
#include <kangaroo/VecMath.h>
#include <kangaroo/Mat.h>
#include <kangaroo/MatUtils.h>
#include <kangaroo/InvalidValue.h>
#include <kangaroo/Image.h>
#include <kangaroo/ImageIntrinsics.h>
#include <kangaroo/BoundingBox.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/Volume.h>
#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Pyramid.h>
#include <kangaroo/launch_utils.h>

#include <kangaroo/cu_operations.h>
#include <kangaroo/cu_resample.h>
#include <kangaroo/reduce.h>
#include <kangaroo/cu_bilateral.h>
#include <kangaroo/cu_depth_tools.h>
#include <kangaroo/cu_normals.h>
#include <kangaroo/cu_sdffusion.h>
#include <kangaroo/cu_raycast.h>
#include <kangaroo/reweighting.h>
#include <kangaroo/cu_model_refinement.h>
#include <kangaroo/MarchingCubes.h>
