// Appended to src/scene/world.rs and src/scene/camera.rs respectively.
use crate::ffi::rt_camera;
use crate::flatten::{camera_block, flatten_world, SceneBuilder};

impl World {
    /// The tables rt_scene_create / rt_multi_create take (include/rt_abi.h).
    pub fn flatten(&self) -> Result<SceneBuilder, String> {
        flatten_world(&self.objects, &self.lights)
    }
}

impl Camera {
    pub fn to_rt_camera(&self) -> Result<rt_camera, String> {
        // width() is vsize and height() is hsize (src/scene/camera.rs:68-74): the canvas `render`
        // fills is width() x height(), while half_width / half_height / pixel_size were derived from
        // hsize / vsize in Camera::new.  Both are passed on as they are.
        camera_block(self.width(), self.height(), self.half_width, self.half_height, self.pixel_size, &self.transform)
    }
}
