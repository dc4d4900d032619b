// Host worker pool of a context and its batches in flight (plain C++: also built alone by tests/emul/pool_test.cpp, under
// ThreadSanitizer).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// Persistent host worker pool.  The host stages of a batch are a few hundred microseconds of work per thread, so creating the
// threads per call (tens of microseconds EACH, serialised) used to cost more than the work.  One pool serves a context AND its
// async lanes (batches in flight): every parallel loop is a job in a shared FIFO, so a lane whose host stage runs alone gets all
// the threads, several lanes' stages share them, and the number of running workers never exceeds the CPUs the process may use
// (a static split -- threads / lanes each -- left most threads idle for long-read batches, whose host stages rarely coincide).
struct GdPool {
	struct Job {
		std::function<void(int)> f;
		int n = 0;
		std::atomic<int> next{0}, done{0};
		std::mutex mu;
		std::condition_variable cv;
	};
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable cv_go;
	std::deque<std::shared_ptr<Job>> jobs;
	bool stop = false;
	static void work_on(Job &J)
	{
		int mine = 0;
		for (;;) {
			const int i = J.next.fetch_add(16);
			if (i >= J.n) break;
			const int e = std::min(J.n, i + 16);
			for (int j = i; j < e; ++j) J.f(j);
			mine += e - i;
		}
		if (mine && J.done.fetch_add(mine) + mine == J.n) { std::unique_lock<std::mutex> lk(J.mu); J.cv.notify_all(); }
	}
	void worker()
	{
		for (;;) {
			std::shared_ptr<Job> J;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_go.wait(lk, [&] { return stop || !jobs.empty(); });
				if (stop) return;
				J = jobs.front();
				if (J->next.load() >= J->n) { jobs.pop_front(); continue; } // handed out completely: the next job's turn
			}
			work_on(*J);
		}
	}
	template <class F> void run(int n_threads, int n_items, F f)
	{
		if (n_threads <= 1 || n_items < 32) { for (int i = 0; i < n_items; ++i) f(i); return; }
		auto J = std::make_shared<Job>();
		J->f = f, J->n = n_items;
		{
			std::unique_lock<std::mutex> lk(mu);
			while ((int)th.size() < n_threads - 1) th.emplace_back([this] { worker(); }); // the caller works too
			jobs.push_back(J);
		}
		cv_go.notify_all();
		work_on(*J);
		{
			std::unique_lock<std::mutex> lk(J->mu);
			J->cv.wait(lk, [&] { return J->done.load() == J->n; });
		}
		std::unique_lock<std::mutex> lk(mu); // (a worker may have dropped it already)
		for (auto it = jobs.begin(); it != jobs.end(); ++it) if (it->get() == J.get()) { jobs.erase(it); break; }
	}
	~GdPool()
	{
		{ std::unique_lock<std::mutex> lk(mu); stop = true; }
		cv_go.notify_all();
		for (auto &t : th) t.join();
	}
};

