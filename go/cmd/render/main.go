// Headless twin of the reference's cmd/render (cmd/render/main.go:14-63) with the UI behind a build
// tag, so the binary needs only libc and libptcore.so.  NOT COMPILED IN THE BUILD IMAGE (no Go).
//
// Reference flags kept verbatim: -scene -mode -gpu -headless -out.
// Additive flags: -width -height -spp -depth (override the mode preset), -seed, -devices,
// -scene-settings (the editor's scene-settings override, internal/ui/app.go:60-75; off = main.go:52).
package main

import (
	"flag"
	"fmt"
	"log"
	"os"

	"github.com/user/pathtracer/internal/engine"
	"github.com/user/pathtracer/internal/engine/hip"
	"github.com/user/pathtracer/internal/scene"
)

func main() {
	log.Println("pathtracer: starting main()")
	scenePath := flag.String("scene", "scenes/example_simple.json", "path to scene JSON file")
	mode := flag.String("mode", "preview", "render mode: preview or final")
	useGPU := flag.Bool("gpu", false, "use GPU backend for rendering (if available)")
	headless := flag.Bool("headless", false, "render without UI and save PNG")
	output := flag.String("out", "output.png", "output PNG file for headless render")
	width := flag.Int("width", 0, "image width (default: the mode preset)")
	height := flag.Int("height", 0, "image height (default: the mode preset)")
	spp := flag.Int("spp", -1, "samples per pixel (default: the mode preset)")
	depth := flag.Int("depth", -1, "max path depth (default: the mode preset)")
	seed := flag.Uint64("seed", 1, "sample-stream seed")
	devices := flag.Int("devices", 1, "number of GPUs to tile the image over")
	sceneSettings := flag.Bool("scene-settings", false, "let the scene file's settings block override the mode preset")
	flag.Parse()
	log.Printf("flags: scene=%s mode=%s headless=%v out=%s\n", *scenePath, *mode, *headless, *output)

	if *useGPU {
		engine.SetBackend(engine.BackendGPU) // RenderInto -> renderIntoGPU -> hip.Render (see INTEGRATION.md)
		hip.SetDevices(*devices)
		hip.SetSeed(*seed)
	} else {
		engine.SetBackend(engine.BackendCPU)
	}
	if !*headless {
		log.Println("ui error: built without the ui tag; use -headless")
		os.Exit(1)
	}
	sc, err := scene.Load(*scenePath)
	if err != nil {
		log.Println("headless render error:", fmt.Errorf("load scene: %w", err))
		os.Exit(1)
	}
	s := engine.RenderSettingsForMode(*mode)
	if *sceneSettings { // internal/ui/app.go:60-75
		if sc.Settings.Width > 0 && sc.Settings.Height > 0 {
			s.Width, s.Height = sc.Settings.Width, sc.Settings.Height
			if sc.Settings.SamplesPerPx > 0 {
				s.SamplesPerPx = sc.Settings.SamplesPerPx
			}
			if sc.Settings.MaxDepth > 0 {
				s.MaxDepth = sc.Settings.MaxDepth
			}
		}
		if *mode == "final" {
			s.SamplesPerPx *= 4
			s.MaxDepth *= 2
		}
	}
	if *width > 0 {
		s.Width = *width
	}
	if *height > 0 {
		s.Height = *height
	}
	if *spp >= 0 {
		s.SamplesPerPx = *spp
	}
	if *depth >= 0 {
		s.MaxDepth = *depth
	}
	img, err := engine.RenderScene(sc, s)
	if err != nil {
		log.Println("headless render error:", fmt.Errorf("render scene: %w", err))
		os.Exit(1)
	}
	if err := engine.SavePNG(*output, img); err != nil {
		log.Println("headless render error:", fmt.Errorf("save png: %w", err))
		os.Exit(1)
	}
}
