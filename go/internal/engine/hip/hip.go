// Package hip is the MI355X backend of the path tracer: a drop-in for package
// internal/engine/gpu (the OpenGL backend) behind engine.RenderInto.
//
//	gpu.Render(sc *scene.Scene, cfg gpu.RenderConfig, img *image.RGBA, progress func()) error
//
// has the same signature here.  The scene is flattened into the plain C structs of
// include/ptcore.h and rendered by libptcore.so (HIP kernels for gfx950).
//
// NOT COMPILED IN THE BUILD IMAGE: no Go toolchain exists there.  The file is the binding a
// maintainer adds to the reference tree (see INTEGRATION.md); it uses only documented cgo rules
// (no Go pointers retained by C, img.Pix is pointer-free and may be passed directly).
package hip

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../../path_trace_golang_amd -lptcore -Wl,-rpath,${SRCDIR}/../../../../path_trace_golang_amd
#include <stdlib.h>
#include "ptcore.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"image"
	"os"
	"runtime"
	"strconv"
	"sync"
	"unsafe"

	"github.com/user/pathtracer/internal/scene"
)

// RenderConfig mirrors gpu.RenderConfig (internal/engine/gpu/gpu.go:227-232).
type RenderConfig struct {
	Width        int
	Height       int
	SamplesPerPx int
	MaxDepth     int
}

var (
	mu      sync.Mutex // one render at a time, like the reference's single GL worker (gpu.go:2534-2546)
	ctx     *C.pt_ctx
	initErr error // sticky, like gpu.go:279-286
	devices = 1
	seed    = uint64(1)
)

// SetDevices selects how many GPUs (ordinals 0..n-1) the frame is tiled over.
// The tiles of a frame are collected on device 0 by peer copies over xGMI, or -- with
// PTCORE_GATHER=rccl in the environment when the context is created -- by grouped RCCL
// send / receive (libptcore loads librccl.so itself in that case; nothing to link here).
func SetDevices(n int) {
	mu.Lock()
	defer mu.Unlock()
	if ctx != nil {
		C.pt_destroy(ctx)
		ctx = nil
	}
	initErr = nil
	if n > 0 {
		devices = n
	}
}

// SetSeed selects the sample streams (the CPU engine seeds from the clock, random.go:14-16).
func SetSeed(s uint64) { seed = s }

func lastError(what string) error {
	return fmt.Errorf("%s: %s", what, C.GoString(C.pt_last_error()))
}

func ensure() error {
	if ctx != nil {
		return nil
	}
	if initErr != nil {
		return initErr
	}
	if s := os.Getenv("PATHTRACER_SEED"); s != "" {
		if v, err := strconv.ParseUint(s, 10, 64); err == nil {
			seed = v
		}
	}
	if rc := C.pt_create(nil, C.int32_t(devices), &ctx); rc != C.PT_OK {
		initErr = lastError("pt_create")
		return initErr
	}
	return nil
}

func matType(t scene.MaterialType) C.int32_t {
	switch t {
	case scene.MaterialMetal:
		return C.PT_MAT_METAL
	case scene.MaterialDielectric:
		return C.PT_MAT_DIELECTRIC
	case scene.MaterialEmissive:
		return C.PT_MAT_EMISSIVE
	case scene.MaterialMirror:
		return C.PT_MAT_MIRROR
	}
	return C.PT_MAT_LAMBERT // convertMaterial's default branch, materials.go:51-53
}

func objType(t scene.ObjectType) C.int32_t {
	switch t {
	case scene.ObjectSphere:
		return C.PT_OBJ_SPHERE
	case scene.ObjectPlane:
		return C.PT_OBJ_PLANE
	case scene.ObjectBox:
		return C.PT_OBJ_BOX
	case scene.ObjectSphereLight:
		return C.PT_OBJ_SPHERE_LIGHT
	}
	return C.PT_OBJ_UNKNOWN // skipped by sceneToWorld, objects.go:237-266
}

func set3(d *[3]C.double, x, y, z float64) { d[0], d[1], d[2] = C.double(x), C.double(y), C.double(z) }

// flatten copies the scene into C memory (freed by the returned func): C never sees a Go pointer
// to memory containing Go pointers.
func flatten(sc *scene.Scene) (*C.pt_scene, func()) {
	nm, no := len(sc.Materials), len(sc.Objects)
	cs := (*C.pt_scene)(C.calloc(1, C.size_t(unsafe.Sizeof(C.pt_scene{}))))
	var mats *C.pt_material
	var objs *C.pt_object
	ids := make(map[string]int, nm)
	if nm > 0 {
		mats = (*C.pt_material)(C.calloc(C.size_t(nm), C.size_t(unsafe.Sizeof(C.pt_material{}))))
		ms := unsafe.Slice(mats, nm)
		for i, m := range sc.Materials {
			ms[i]._type = matType(m.Type)
			set3(&ms[i].albedo, m.Albedo.R, m.Albedo.G, m.Albedo.B)
			ms[i].rough = C.double(m.Rough)
			ms[i].ior = C.double(m.IOR)
			set3(&ms[i].emit, m.Emit.R, m.Emit.G, m.Emit.B)
			ms[i].power = C.double(m.Power)
			set3(&ms[i].absorption, m.Absorption.R, m.Absorption.G, m.Absorption.B)
			ms[i].smoothness = C.double(m.Smoothness)
			ids[m.ID] = i // the last duplicate wins, like the map at objects.go:227-229
		}
	}
	if no > 0 {
		objs = (*C.pt_object)(C.calloc(C.size_t(no), C.size_t(unsafe.Sizeof(C.pt_object{}))))
		os_ := unsafe.Slice(objs, no)
		for i, o := range sc.Objects {
			os_[i]._type = objType(o.Type)
			if k, ok := ids[o.MaterialID]; ok {
				os_[i].material = C.int32_t(k)
			} else {
				os_[i].material = -1
			}
			set3(&os_[i].position, o.Position.X, o.Position.Y, o.Position.Z)
			set3(&os_[i].size, o.Size.X, o.Size.Y, o.Size.Z)
		}
	}
	c := sc.Camera
	set3(&cs.camera.position, c.Position.X, c.Position.Y, c.Position.Z)
	set3(&cs.camera.target, c.Target.X, c.Target.Y, c.Target.Z)
	set3(&cs.camera.up, c.Up.X, c.Up.Y, c.Up.Z)
	cs.camera.fov = C.double(c.FOV)
	cs.camera.aperture = C.double(c.Aperture)
	cs.camera.focus_dist = C.double(c.FocusDist)
	cs.camera.aspect_ratio = C.double(c.AspectRatio)
	set3(&cs.sky.background, sc.Background.R, sc.Background.G, sc.Background.B)
	cs.sky.kind = C.PT_SKY_BACKGROUND
	if sc.Sky != nil {
		switch sc.Sky.Type {
		case "gradient":
			cs.sky.kind = C.PT_SKY_GRADIENT
		case "solid":
			cs.sky.kind = C.PT_SKY_SOLID
		}
		set3(&cs.sky.color, sc.Sky.Color.R, sc.Sky.Color.G, sc.Sky.Color.B)
		set3(&cs.sky.horizon, sc.Sky.Horizon.R, sc.Sky.Horizon.G, sc.Sky.Horizon.B)
		set3(&cs.sky.zenith, sc.Sky.Zenith.R, sc.Sky.Zenith.G, sc.Sky.Zenith.B)
	}
	cs.num_materials = C.int32_t(nm)
	cs.num_objects = C.int32_t(no)
	cs.materials = mats
	cs.objects = objs
	return cs, func() {
		C.free(unsafe.Pointer(mats))
		C.free(unsafe.Pointer(objs))
		C.free(unsafe.Pointer(cs))
	}
}

// Render renders sc into img on the MI355X and calls progress() every ~10% of the samples and once
// at the end (the cadence of gpu.go:2209-2212, :2229, :2523-2525).  On any error the caller
// (engine.renderIntoGPU) falls back to the CPU renderer exactly as it does for the GL backend.
func Render(sc *scene.Scene, cfg RenderConfig, img *image.RGBA, progress func()) error {
	if sc == nil || img == nil {
		return errors.New("hip.Render: nil scene or image")
	}
	b := img.Bounds()
	if b.Dx() != cfg.Width || b.Dy() != cfg.Height {
		return nil // renderIntoCPU silently returns on a size mismatch (renderer.go:46-49)
	}
	if len(img.Pix) == 0 {
		return errors.New("hip.Render: empty image")
	}
	mu.Lock()
	defer mu.Unlock()
	// pt_last_error() is thread-local on the C side: stay on one OS thread from a failing call to its message
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if err := ensure(); err != nil {
		return err
	}
	cs, free := flatten(sc)
	defer free()
	pc := C.pt_config{width: C.int32_t(cfg.Width), height: C.int32_t(cfg.Height),
		samples_per_px: C.int32_t(cfg.SamplesPerPx), max_depth: C.int32_t(cfg.MaxDepth), seed: C.uint64_t(seed)}
	pix := (*C.uint8_t)(unsafe.Pointer(&img.Pix[0]))
	if progress == nil {
		if rc := C.pt_render(ctx, cs, &pc, pix, C.int32_t(img.Stride), nil, nil, nil, nil); rc != C.PT_OK {
			return lastError("pt_render")
		}
		return nil
	}
	if rc := C.pt_begin(ctx, cs, &pc); rc != C.PT_OK {
		return lastError("pt_begin")
	}
	step := cfg.SamplesPerPx / 10
	if step < 1 {
		step = 1
	}
	var done C.int32_t
	var err error
	for int(done) < cfg.SamplesPerPx {
		if rc := C.pt_step(ctx, C.int32_t(step), &done); rc != C.PT_OK {
			err = lastError("pt_step")
			break
		}
		if rc := C.pt_read(ctx, pix, C.int32_t(img.Stride), nil); rc != C.PT_OK {
			err = lastError("pt_read")
			break
		}
		progress()
	}
	if err == nil && cfg.SamplesPerPx <= 0 {
		// zero samples: the reference's pixel finish of an empty sum (renderer.go:190-221)
		if rc := C.pt_read(ctx, pix, C.int32_t(img.Stride), nil); rc != C.PT_OK {
			err = lastError("pt_read")
		}
	}
	if rc := C.pt_end(ctx, nil); rc != C.PT_OK && err == nil {
		err = lastError("pt_end")
	}
	if err == nil {
		progress()
	}
	return err
}
