// PIN KIT — numbers that decide the oracle's reading of the un-vendored `math` (rust_cg_math) and `rust_optics` crates.
//
// This file is NOT part of the MI355X engine and was written WITHOUT a Rust toolchain: it is UNVERIFIED Rust, modelled line by line on calls
// the reference tree itself makes (src/parsing/curves.rs:405-477, src/materials/ggx.rs:630-660, src/parsing/instance.rs:40-70,
// src/integrator/pt.rs:240,353,501).  A maintainer who has `cargo` and the git crates:
//
//   1. copies this file to  <reference>/src/pin_kit.rs  and adds  `#[cfg(test)] mod pin_kit;`  to  <reference>/src/lib.rs
//   2. runs   cargo test --release pin_kit -- --nocapture | grep '^PIN ' > pin.txt
//   3. runs   python tools/pin_kit/pin_to_json.py pin.txt > tests/golden/reference_pin.json     (in this repository)
//   4. runs   python -m pytest tests/test_pin.py -q
//
// Every line it prints is `PIN <key> <f32 bit pattern, hex> <decimal>`.  tests/test_pin.py evaluates the same expressions on the oracle
// (oracle/ptref.cpp) and compares; the table in DESIGN.md section 2 says which key decides which restated choice.  If a call below does not
// compile against the crate (a renamed method, a tuple where a struct is expected), fix the call, not the key: the keys are the contract.
#![allow(unused_imports)]
use crate::curves;
use crate::materials::{Material, GGX};
use crate::prelude::*;
use math::curves::InterpolationMode;

fn pin(key: &str, v: f32) {
    println!("PIN {} {:08x} {:e}", key, v.to_bits(), v);
}
fn pin3(key: &str, v: Vec3) {
    pin(&format!("{}.x", key), v.x());
    pin(&format!("{}.y", key), v.y());
    pin(&format!("{}.z", key), v.z());
}

#[test]
fn pin_kit() {
    // ---- Curve::Tabulated, InterpolationMode::Cubic (zero-tangent Hermite? Catmull-Rom? natural spline?) and its behaviour outside the knots
    let tab = Curve::Tabulated {
        signal: vec![(400.0, 0.1), (450.0, 0.5), (520.0, 0.3), (600.0, 0.9), (700.0, 0.2)],
        mode: InterpolationMode::Cubic,
    };
    for (i, l) in [380.0f32, 425.0, 500.0, 560.5, 650.0, 720.0].iter().enumerate() {
        pin(&format!("curve.tabulated.cubic.{}", i), tab.evaluate(*l));
    }
    let tab_lin = Curve::Tabulated {
        signal: vec![(400.0, 0.1), (450.0, 0.5), (520.0, 0.3), (600.0, 0.9), (700.0, 0.2)],
        mode: InterpolationMode::Linear,
    };
    pin("curve.tabulated.linear.0", tab_lin.evaluate(500.0));
    // ---- Curve::Linear: outside the bounds, the last bin, the upper bound itself (the oracle clamps the index there: DESIGN.md section 10)
    let lin = Curve::Linear {
        signal: vec![0.2, 0.8, 0.4, 1.0],
        bounds: Bounds1D::new(400.0, 600.0),
        mode: InterpolationMode::Cubic,
    };
    for (i, l) in [390.0f32, 425.0, 475.0, 560.0, 599.0, 610.0].iter().enumerate() {
        pin(&format!("curve.linear.cubic.{}", i), lin.evaluate(*l));
    }
    // ---- evaluate_clamped (DiffuseLight's bounce colour): clamped to [0, 1]?
    let big = Curve::Linear { signal: vec![1.5], bounds: EXTENDED_VISIBLE_RANGE, mode: InterpolationMode::Linear };
    pin("curve.evaluate_clamped.0", big.evaluate_clamped(550.0));
    pin("curve.evaluate_power.0", big.evaluate_power(550.0));
    // ---- Blackbody: SI Planck, boost == 0 unnormalised, else boost * B(l) / B(Wien peak)
    pin("curve.blackbody.5000.boost1.550", curves::blackbody_curve(5000.0, 1.0).evaluate(550.0));
    pin("curve.blackbody.5000.boost1.400", curves::blackbody_curve(5000.0, 1.0).evaluate(400.0));
    pin("curve.blackbody.3000.boost5.650", curves::blackbody_curve(3000.0, 5.0).evaluate(650.0));
    pin("curve.blackbody.5000.boost0.550", curves::blackbody_curve(5000.0, 0.0).evaluate(550.0));
    // ---- Cauchy, Exponential (the mauve error light)
    pin("curve.cauchy.400", curves::cauchy(1.4, 30000.0).evaluate(400.0));
    pin("curve.cauchy.700", curves::cauchy(1.4, 30000.0).evaluate(700.0));
    pin("curve.exponential.mauve.500", curves::mauve(1.0).evaluate(500.0));
    pin("curve.exponential.mauve.650", curves::mauve(1.0).evaluate(650.0));
    // ---- XYZColor::from(SingleWavelength): energy * (x_bar, y_bar, z_bar), multi-lobe fit evaluated in f64 at Angstrom?
    for l in [450.0f32, 550.0, 650.0].iter() {
        let c: XYZColor = SingleWavelength::new(*l, 1.0).into();
        let [x, y, z, _]: [f32; 4] = c.0.into();
        pin(&format!("xyz.{}.x", *l as u32), x);
        pin(&format!("xyz.{}.y", *l as u32), y);
        pin(&format!("xyz.{}.z", *l as u32), z);
    }
    // ---- uv <-> direction: polar axis (+Z?), where u = 0 points
    pin3("uv_to_direction.0", uv_to_direction((0.25, 0.5)));
    pin3("uv_to_direction.1", uv_to_direction((0.7, 0.2)));
    pin3("uv_to_direction.2", uv_to_direction((0.5, 0.0)));
    let (u, v) = direction_to_uv(Vec3::new(0.3, -0.5, 0.8).normalized());
    pin("direction_to_uv.0.u", u);
    pin("direction_to_uv.0.v", v);
    // ---- MIS heuristics: math::power_heuristic (beta = 2?) and the in-tree power_heuristic_generic (src/lib.rs:114-119)
    pin("power_heuristic.0", power_heuristic(0.7, 0.2));
    pin("power_heuristic_generic.0", crate::power_heuristic_generic(0.7f32, 0.2f32));
    // ---- sampling helpers
    pin3("random_cosine_direction.0", random_cosine_direction(Sample2D { x: 0.3, y: 0.6 }));
    pin3("random_cosine_direction.1", random_cosine_direction(Sample2D { x: 0.9, y: 0.1 }));
    pin3("random_on_unit_sphere.0", random_on_unit_sphere(Sample2D { x: 0.3, y: 0.6 }));
    pin3("random_in_unit_disk.0", random_in_unit_disk(Sample2D { x: 0.3, y: 0.6 }));
    // ---- TangentFrame::from_normal (Duff et al. 2017? Frisvad?)
    let n = Vec3::new(0.3, -0.5, 0.8).normalized();
    let frame = TangentFrame::from_normal(n);
    pin3("frame.to_world.0", frame.to_world(&Vec3::new(0.2, 0.4, 0.7)));
    pin3("frame.to_local.0", frame.to_local(&Vec3::new(0.2, 0.4, 0.7)));
    let frame_down = TangentFrame::from_normal(Vec3::new(0.1, 0.2, -0.97).normalized());
    pin3("frame.to_world.1", frame_down.to_world(&Vec3::new(0.2, 0.4, 0.7)));
    // ---- Sample1D::choose: `<` or `<=`, is the sample rescaled
    let (s0, c0) = Sample1D { x: 0.3 }.choose(0.4, 1.0f32, 2.0f32);
    pin("choose.0.x", s0.x);
    pin("choose.0.choice", c0);
    let (s1, c1) = Sample1D { x: 0.7 }.choose(0.4, 1.0f32, 2.0f32);
    pin("choose.1.x", s1.x);
    pin("choose.1.choice", c1);
    let (s2, c2) = Sample1D { x: 0.4 }.choose(0.4, 1.0f32, 2.0f32);
    pin("choose.2.x", s2.x);
    pin("choose.2.choice", c2);
    // ---- Transform3::from_stack (scale, rotate about z by 90 degrees, translate): order of composition, f32 or f64
    let t = Transform3::from_stack(
        Some(Transform3::from_scale(Vec3::new(0.9, 0.9, 0.9))),
        Some(Transform3::from_axis_angle(Vec3::new(0.0, 0.0, 1.0).normalized(), PI * 90.0 / 180.0)),
        Some(Transform3::from_translation(Vec3::new(0.0, 0.0, -0.1))),
    );
    let p = t.to_world(Point3::new(0.5, 0.25, 1.0));
    pin("transform.point.x", p.x());
    pin("transform.point.y", p.y());
    pin("transform.point.z", p.z());
    // ---- GGX (in-tree code on top of the crate's curves and vectors): the proptest regression seed (proptest-regressions/materials/ggx.txt:7)
    // and one ordinary sample; glass as in src/materials/ggx.rs:630-635
    let glass = |roughness: f32| GGX::new(roughness, curves::cauchy(1.5, 10000.0), curves::cie_e(1.0), curves::void(), 0, 0);
    {
        let m = glass(8.736748);
        let wi = Vec3::new(0.54826164, 0.0, -0.83630687);
        let (f, wo, pdf) = m.generate_and_evaluate(400.0, UV(0.0, 0.0), TransportMode::Importance, Sample2D { x: 0.0, y: 0.0 }, wi);
        pin("ggx.seed.f", f);
        pin("ggx.seed.pdf", *pdf);
        pin3("ggx.seed.wo", wo.unwrap());
    }
    {
        let m = glass(0.2);
        let wi = Vec3::new(0.3, 0.2, 0.93).normalized();
        let (f, wo, pdf) = m.generate_and_evaluate(550.0, UV(0.0, 0.0), TransportMode::Importance, Sample2D { x: 0.3, y: 0.7 }, wi);
        pin("ggx.rough.f", f);
        pin("ggx.rough.pdf", *pdf);
        pin3("ggx.rough.wo", wo.unwrap());
        let wo2 = Vec3::new(-0.2, 0.1, -0.97).normalized();
        let (f2, pdf2) = m.bsdf(550.0, UV(0.0, 0.0), TransportMode::Importance, wi, wo2);
        pin("ggx.rough.bsdf.f", f2);
        pin("ggx.rough.bsdf.pdf", *pdf2);
    }
}
