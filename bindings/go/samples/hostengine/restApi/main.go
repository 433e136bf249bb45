// restApi serves GPU information over HTTP on :8070 from an embedded host
// engine: the routes of the reference's samples/dcgm/restApi (server.go:40-71),
// under /dcgm/... for compatibility and /gpumon/... as the neutral prefix.
//
//	/dcgm/device/info/{id|uuid}/{value}[/json]
//	/dcgm/device/status/{id|uuid}/{value}[/json]
//	/dcgm/process/info/pid/{pid}[/json]
//	/dcgm/health/{id|uuid}/{value}[/json]
//	/dcgm/status[/json]
//
// The gpumon-hostengine daemon serves the same routes natively (--http); this
// sample shows how to build them on the Go API with only the standard library.
package main

import (
	"context"
	"flag"
	"log"
	"net/http"
	"os"
	"os/signal"
	"syscall"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

const timeout = 5 * time.Second

func main() {
	addr := flag.String("listen", ":8070", "address to serve on")
	flag.Parse()
	cleanup, err := he.Init(he.Embedded)
	if err != nil {
		log.Panicln("cannot start the host engine:", err)
	}
	defer cleanup()
	api, err := newAPI()
	if err != nil {
		log.Panicln(err)
	}
	srv := &http.Server{Addr: *addr, Handler: api, ReadTimeout: timeout, WriteTimeout: 2 * timeout}
	stop := make(chan os.Signal, 1)
	signal.Notify(stop, os.Interrupt, syscall.SIGTERM)
	go func() {
		log.Printf("Running http server on %s", *addr)
		if err := srv.ListenAndServe(); err != http.ErrServerClosed {
			log.Panicln(err)
		}
	}()
	<-stop
	ctx, cancel := context.WithTimeout(context.Background(), timeout)
	defer cancel()
	if err := srv.Shutdown(ctx); err != nil {
		log.Println("shutdown:", err)
	} else {
		log.Println("http server stopped")
	}
}
