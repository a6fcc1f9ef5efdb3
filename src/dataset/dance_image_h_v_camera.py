"""src.dataset.dance_image_h_v_camera: only the inference-time camera front-end (Camera, ray_condition);
the training datasets / samplers of the reference are out of scope."""
from humanvid_amd.camera import Camera, camera_file_to_embedding, get_relative_pose, load_cameras, ray_condition  # noqa: F401
