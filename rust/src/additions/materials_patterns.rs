// Appended to src/materials/phong.rs and src/patterns/{solid,striped,gradient,ring,checker,blended,
// perturbed,test_pattern}.rs respectively.
use crate::ffi::{RT_PAT_BLENDED, RT_PAT_CHECKER, RT_PAT_GRADIENT, RT_PAT_RING, RT_PAT_STRIPED};
use crate::flatten::{FlatMaterial, FlatPattern};

impl Phong {
    pub(crate) fn flat_node(&self) -> FlatMaterial<'_> {
        FlatMaterial {
            pattern: self.pattern.as_ref(),
            ambient: self.ambient,
            diffuse: self.diffuse,
            specular: self.specular,
            shininess: self.shininess,
            reflective: self.reflective,
            transparency: self.transparency,
            refractive_index: self.refractive_index,
        }
    }
}

impl Solid {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Solid { color: self.color }
    }
}

impl Striped {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Binary { kind: RT_PAT_STRIPED, former: self.former.as_ref(), latter: self.latter.as_ref() }
    }
}

impl Gradient {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Binary { kind: RT_PAT_GRADIENT, former: self.former.as_ref(), latter: self.latter.as_ref() }
    }
}

impl Ring {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Binary { kind: RT_PAT_RING, former: self.former.as_ref(), latter: self.latter.as_ref() }
    }
}

impl Checker {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Binary { kind: RT_PAT_CHECKER, former: self.former.as_ref(), latter: self.latter.as_ref() }
    }
}

impl Blended {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Binary { kind: RT_PAT_BLENDED, former: self.former.as_ref(), latter: self.latter.as_ref() }
    }
}

impl Perturbed {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Perturbed { delegate: self.delegate.as_ref(), scale: self.scale }
    }
}

impl TestPattern {
    pub(crate) fn flat_node(&self) -> FlatPattern<'_> {
        FlatPattern::Test
    }
}
