// Appended to src/geometry/{sphere,plane,cube,cylinder,cone,triangle,smooth_triangle,group,csg,
// test_shape}.rs respectively (each block to its own file).
use crate::flatten::FlatShape;

impl Sphere {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Sphere
    }
}

impl Plane {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Plane
    }
}

impl Cube {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Cube
    }
}

impl Cylinder {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Cylinder { minimum: self.minimum, maximum: self.maximum, closed: self.closed }
    }
}

impl Cone {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Cone { minimum: self.minimum, maximum: self.maximum, closed: self.closed }
    }
}

impl Triangle {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Triangle { p1: self.p1, e1: self.e1, e2: self.e2, normal: self.normal }
    }
}

impl SmoothTriangle {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::SmoothTriangle { p1: self.p1, e1: self.e1, e2: self.e2, n1: self.n1, n2: self.n2, n3: self.n3 }
    }
}

impl Group {
    pub(crate) fn flat_node(&self) -> FlatShape {
        FlatShape::Group { children: self.children.read().unwrap().clone() }
    }
}

impl CSG {
    pub(crate) fn flat_node(&self) -> FlatShape {
        let operation = match self.operation {
            Operation::Difference => 0,
            Operation::Intersection => 1,
            Operation::Union => 2,
        };
        FlatShape::Csg { operation, left: self.left.clone(), right: self.right.clone() }
    }
}

impl TestShape {
    // a test double (src/geometry/test_shape.rs): it records the ray it is given and has no place on the device
    pub(crate) fn flat_node(&self) -> FlatShape {
        panic!("TestShape cannot be rendered on the GPU")
    }
}
