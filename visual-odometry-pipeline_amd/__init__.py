"""MI355X (gfx950) implementation of the per-frame hot path of the MATLAB visual-odometry
pipeline, behind the reference's own function names.  All arithmetic runs in lib/libvohip.so
(hand-written HIP); this package is the thin host layer above its C ABI (include/vohip.h).
"""
from . import _lib
from ._lib import Context, VohipError, default_context
from . import batch, datasets, synth
from .datasets import dataset_config, eval_sequence, load_sequence, run_dataset
from .ba import T_refinement, T_refinement_batch, baError, runBA
from .bootstrap import (bootstrap, compProjMat, estimateEssentialMatrix, matching_init, pose_estimation_init,
                        relativeCameraPose, triangulation_init)
from .evaluation import (MAIN_EVAL_FEATS, alignError, alignEstimateToGroundTruth, experiment_name, load_results, main_eval,
                         quantitative_eval, run_group, run_sequence, save_results)
from .fmatrix import draw_octets, estimateFundamentalMatrix, octets_from_uniforms
from .frame_pipeline import VOGroupDevice, VOPipelineDevice
from .keypoints import BRISKPoints, SURFPoints, cornerPoints, extractFeatures, genKeypoints, matchFeatures
from .klt import PointTracker, Pyramid
from .pipeline import (VO_bundle_adjust, VOPipeline, default_hyper_paras, init_state, update_candidate,
                       update_landmarks)
from .ransac import (draw_sextets, draw_triples, estimatePoseDLT, p3p, ransacLocalization, sets_from_uniforms,
                     triples_from_uniforms)
from .geometry import (Cross2Matrix, HomogMatrix2twist, Matrix2Cross, candidate_filter, candidate_triangulate,
                       linearTriangulation, linearTriangulation_refinement, twist2HomogMatrix)

__all__ = ["Context", "VohipError", "default_context", "ransacLocalization", "p3p", "draw_triples"]
