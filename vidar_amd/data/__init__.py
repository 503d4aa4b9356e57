"""The data format on the input side of the hot path: `assemble` turns per-frame
records into the sample the detector consumes, `reader` produces those records from a nuScenes / OpenScene
info pkl (images, multi-sweep lidar, voxel subsampling)."""
from .assemble import (frame_index_lists, frame_meta_from_info, transform_matrix, union2one,
                       usable_indices)  # noqa: F401
from .augment import CropResizeFlipImage, PhotoMetricDistortionMultiViewImage  # noqa: F401
from .reader import (TrainAugment, ViDARSequenceDataset, load_images, load_infos, load_multi_sweeps, load_pcd_file, load_points_file,  # noqa: F401
                     voxel_subsample)
from .loader import DistributedGroupSampler, DistributedSampler, build_dataloader, collate  # noqa: F401
