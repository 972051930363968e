// ---- src/geometry/shape.rs: one more required method of `trait Shape` ------------------------
//     fn flat(&self) -> crate::flatten::FlatShape;
// ---- src/materials/material.rs: one more required method of `trait Material` -----------------
//     fn flat(&self) -> crate::flatten::FlatMaterial<'_>;
// ---- src/patterns/pattern.rs: one more required method of `trait Pattern` --------------------
//     fn flat(&self) -> crate::flatten::FlatPattern<'_>;
//
// and in every `impl Shape for X` / `impl Material for Phong` / `impl Pattern for X` block:
//     fn flat(&self) -> ... { self.flat_node() }
// where `flat_node` is the inherent method in the files next to this one.
